// Microbenchmark: cost of a cooperative-groups grid barrier on MI355X with the render
// kernels' launch shape (one 1024-thread workgroup per CU), against the ~4.8 us that a
// dependent-kernel boundary costs.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/grid_sync.hip -o /tmp/grid_sync && /tmp/grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdio.h>
namespace cg = cooperative_groups;
__global__ __launch_bounds__(1024) void k_sync(float* out, int n) {
  cg::grid_group g = cg::this_grid();
  float x = threadIdx.x;
  for (int i = 0; i < n; ++i) {
    x = x * 1.0001f + 1.f;
    g.sync();
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ __launch_bounds__(1024) void k_empty(float* out) { out[blockIdx.x * blockDim.x + threadIdx.x] = 1.f; }
int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 1024 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int n : {0, 1, 2, 8, 32}) {
    float ms = 0, best = 1e9;
    void* args[] = {&out, &n};
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0);
      hipError_t e = hipLaunchCooperativeKernel((void*)k_sync, dim3(256), dim3(1024), args, 0, 0);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
      if (ms < best) best = ms;
    }
    printf("cooperative kernel with %2d grid syncs: %.2f us\n", n, best * 1e3);
  }
  {
    float ms = 0, best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0);
      for (int i = 0; i < 32; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(1024), 0, 0, out);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("32 dependent empty kernels: %.2f us each\n", best * 1e3 / 32);
  }
  return 0;
}
