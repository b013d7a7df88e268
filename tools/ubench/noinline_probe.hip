// Which pieces of "a noinline device function called from a kernel" work on gfx950/ROCm 7.2: reading the caller's
// kernarg segment (__builtin_amdgcn_kernarg_segment_ptr in a non-kernel function), generic pointers to LDS, a struct
// returned by value.  Each piece is its own kernel so that a fault names its cause.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args { int n; float scale; const float* in; float* out; int pad[20]; };
typedef const __attribute__((address_space(4))) Args* cargs;
struct Sums { float v[22]; int flag; };

// (__builtin_amdgcn_kernarg_segment_ptr() inside the callee is NULL on this toolchain -- measured: memory fault at
// address 0 -- so the kernel hands its own kernarg pointer over as an ordinary argument)
__device__ __attribute__((noinline)) float f_kernarg(int i, unsigned long long kp) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)kp), hi = __builtin_amdgcn_readfirstlane((unsigned)(kp >> 32));
  cargs p = (cargs)(((unsigned long long)hi << 32) | lo);
  return p->in[i] * p->scale + (float)p->n;
}
__global__ void k_kernarg(Args a) {
  int i = threadIdx.x;
  cargs p = (cargs)__builtin_amdgcn_kernarg_segment_ptr();
  a.out[i] = f_kernarg(i, (unsigned long long)p);
}

__device__ __attribute__((noinline)) float f_lds(const float* lds, int i) { return lds[i] + lds[(i + 1) & 63]; }
__global__ void k_lds(Args a) {
  __shared__ float s[64];
  int i = threadIdx.x;
  s[i] = a.in[i];
  __syncthreads();
  a.out[i] = f_lds(s, i);
}

__device__ __attribute__((noinline)) Sums f_struct(int i, const float* lds) {
  Sums r;
  for (int n = 0; n < 22; ++n) r.v[n] = lds[(i + n) & 63] * (float)(n + 1);
  r.flag = i & 1;
  return r;
}
__global__ void k_struct(Args a) {
  __shared__ float s[64];
  int i = threadIdx.x;
  s[i] = a.in[i];
  __syncthreads();
  Sums r = f_struct(i, s);
  float t = 0.f;
  for (int n = 0; n < 22; ++n) t += r.v[n];
  a.out[i] = t + (float)r.flag;
}

int main() {
  float h_in[64], h_out[64], *d_in, *d_out;
  for (int i = 0; i < 64; ++i) h_in[i] = (float)i * 0.5f;
  CK(hipMalloc(&d_in, 256)); CK(hipMalloc(&d_out, 256));
  CK(hipMemcpy(d_in, h_in, 256, hipMemcpyHostToDevice));
  Args a = {}; a.n = 3; a.scale = 2.f; a.in = d_in; a.out = d_out;
  hipLaunchKernelGGL(k_kernarg, dim3(1), dim3(64), 0, 0, a);
  CK(hipDeviceSynchronize()); CK(hipMemcpy(h_out, d_out, 256, hipMemcpyDeviceToHost));
  printf("kernarg in callee: out[5] = %g (expect %g)\n", h_out[5], h_in[5] * 2.f + 3.f); fflush(stdout);
  hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, a);
  CK(hipDeviceSynchronize()); CK(hipMemcpy(h_out, d_out, 256, hipMemcpyDeviceToHost));
  printf("generic LDS pointer in callee: out[5] = %g (expect %g)\n", h_out[5], h_in[5] + h_in[6]); fflush(stdout);
  hipLaunchKernelGGL(k_struct, dim3(1), dim3(64), 0, 0, a);
  CK(hipDeviceSynchronize()); CK(hipMemcpy(h_out, d_out, 256, hipMemcpyDeviceToHost));
  float e = 0.f; for (int n = 0; n < 22; ++n) e += h_in[(5 + n) & 63] * (float)(n + 1);
  printf("struct by value from callee: out[5] = %g (expect %g)\n", h_out[5], e + 1.f); fflush(stdout);
  return 0;
}
