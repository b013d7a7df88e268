// Microbenchmark: how fast can one SIMD run rrt::adjoint_ray (the k_bwd inner
// body: Phong+specular sphere) from registers, at 1..8 waves per SIMD?  This is
// the compute roofline of k_bwd for its real instruction mix (dependent chains,
// 11 transcendentals per ray), to compare with the achieved per-row time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "rrt_device.h"
using namespace rrt;
constexpr int ITERS = 512;
__global__ void k_adj(const float* ps_, float* out, int spec) {
  typedef const __attribute__((address_space(4))) float* cfptr;
  cfptr q = (cfptr)ps_;
  float A[9] = {q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8]};
  float o[3] = {q[9], q[10], q[11]};
  float colI[3] = {q[13], q[14], q[15]};
  float ks = q[16], kd = q[17], ka = q[18], sh = q[19];
  float L[3] = {q[24], q[25], q[26]}, look[3] = {q[28], q[29], q[30]};
  int t = threadIdx.x + blockIdx.x * blockDim.x;
  float acc[ACC_USED];
  for (int n = 0; n < ACC_USED; ++n) { acc[n] = 0.f; asm volatile("" : "+v"(acc[n])); }
  float u = (threadIdx.x & 63) * 1e-4f - 0.003f;
  for (int it = 0; it < ITERS; ++it) {
    float w = it * 1e-5f - 0.002f;
    float inv = __builtin_amdgcn_rsqf(u * u + w * w + 1.f);
    float d[3] = {u * inv, w * inv, inv};
    float up[3] = {0.3f + w, 0.2f, -0.1f + u};
    adjoint_ray<float>(KIND_SPHERE, A, o, colI, ks, kd, ka, sh, POW_INT, SHADER_PHONG, spec, 1.f, L, look, d, up, acc);
  }
  float s = 0;
  for (int n = 0; n < ACC_USED; ++n) s += acc[n];
  out[t] = s;
}
int main() {
  float h[32] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, -4.f, 15.f, .5f, .6f, .7f, .3f, .7f, .5f, 50.f, 0, 0, 0, 0,
                 -.4f, -.4f, .82f, 0, 0, 0, 1.f, 0};
  float *ps, *out;
  hipMalloc(&ps, sizeof(h)); hipMemcpy(ps, h, sizeof(h), hipMemcpyHostToDevice);
  hipMalloc(&out, 256 * 8 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc = 1; wpc <= 8; ++wpc) {
    if (wpc == 7) continue;
    int blocks = 256 * wpc;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_adj, dim3(blocks), dim3(256), 0, 0, ps, out, 1);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    double calls = (double)blocks * 4 * ITERS;       // wave-level adjoint calls (64 rays each)
    printf("waves/SIMD %d: %.1f ns per 64-ray adjoint per SIMD  (%.2f Grays/s chip-wide)\n", wpc,
           ms * 1e6 / (calls / 1024), calls * 64 / ms / 1e6);
  }
  return 0;
}
