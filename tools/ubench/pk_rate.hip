// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950.
// Decides whether packing two rays per lane (float2 math) can beat the
// scalar-per-lane kernels (MI355X FP32 vector peak is quoted with packed math).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;
__global__ void k_scalar(float* out, float a, float b) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_packed(float* out, float a, float b) {
  f2 x[8];
  for (int i = 0; i < 8; ++i) { x[i].x = threadIdx.x * 0.001f + i; x[i].y = x[i].x + 0.5f; }
  f2 av = {a, a}, bv = {b, b};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_trans(float* out, float a) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i + 1.f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_rcpf(x[i]) + a;
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc = 1; wpc <= 8; wpc *= 2) {       // workgroups (of 256 threads) per CU => waves per SIMD
    int blocks = 256 * wpc;
    float ms[3];
    for (int k = 0; k < 3; ++k) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (k == 0) hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        if (k == 1) hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        if (k == 2) hipLaunchKernelGGL(k_trans, dim3(blocks), dim3(256), 0, 0, out, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[k], e0, e1);
      }
    }
    double inst = (double)blocks * 4 * ITERS * 8;    // wave-instructions
    printf("waves/SIMD %d: v_fma %.2f ns/wave-instr/SIMD (%.1f TFLOP/s)  v_pk_fma %.2f (%.1f TFLOP/s)  v_rcp+add %.2f\n", wpc,
           ms[0] * 1e6 / (inst / 1024), inst * 128 / ms[0] / 1e9, ms[1] * 1e6 / (inst / 1024), inst * 256 / ms[1] / 1e9,
           ms[2] * 1e6 / (inst / 1024));
  }
  return 0;
}
