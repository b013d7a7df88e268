// store_patterns.hip -- how fast can 256 workgroups write a 4096 x 4096 x 12 B row-major image (201 MB, row stride
// 49152 B), and does the ORDER in which tiles are written matter?  (k_fwd writes exactly this image: 75 % of it zeros.)
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/store_patterns.hip -o build/ubench/store_patterns
//
// Every store is a genuine global_store_dwordx3 (inline asm: left to the compiler, a constant 12-byte store may come out
// as three dword stores).  One 1024-thread workgroup per CU; a wave writes units of `tw` 64-pixel tiles side by side
// (tw * 768 contiguous bytes per row) over 4 rows.
//   mode 0  slab:        workgroup b writes the contiguous 768 KB [b * 768 KB, (b + 1) * 768 KB)          (reference)
//   mode 1  region:      workgroup = a 256 x 256-pixel region of the image, its units in order
//   mode 2  interleaved: the image's units in linear order, unit u belongs to workgroup u mod 256 (k_fwd's deal)
//   mode 4 / 5 / 6  interleaved, one tile per unit, stores with sc1 / sc0 sc1 / nt
// Prints the time per launch (HIP events over 20 launches) for each.
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float v3f __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void st3(char* p, v3f v) {
  asm volatile("global_store_dwordx3 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
// cache-policy variants of the same store (mode 4: sc1, 5: sc0 sc1, 6: nt)
__device__ __forceinline__ void st3_sc1(char* p, v3f v) { asm volatile("global_store_dwordx3 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st3_sc01(char* p, v3f v) { asm volatile("global_store_dwordx3 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st3_nt(char* p, v3f v) { asm volatile("global_store_dwordx3 %0, %1, off nt" ::"v"(p), "v"(v) : "memory"); }

__global__ __launch_bounds__(1024) void k_pattern(float* buf, int mode, int tw, int fma, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const v3f zero = {0.f, 0.f, 0.f};
  char* img = reinterpret_cast<char*>(buf);
  if (mode == 0) {
    char* dst = img + (size_t)blockIdx.x * 786432 + (size_t)threadIdx.x * 12;
    for (int i = 0; i < 64; ++i) st3(dst + (size_t)i * 12288, zero);
    return;
  }
  const int ux_n = 64 / tw, n_units = ux_n * 1024, per_wg = n_units / 256;
  float x = (float)lane;
  for (int i = wave; i < per_wg; i += 16) {
    int ux, uy;
    if (mode >= 4) {            // interleaved tiles (as mode 2, one tile per unit) with a cache policy on the stores
      const int u = i * 256 + (int)blockIdx.x;
      char* base = img + ((size_t)(u / 64) * 4 * 4096 + (size_t)(u % 64) * 64) * 12 + lane * 12;
      for (int p = 0; p < 4; ++p) {
        if (mode == 4) st3_sc1(base + (size_t)p * 49152, zero);
        else if (mode == 5) st3_sc01(base + (size_t)p * 49152, zero);
        else st3_nt(base + (size_t)p * 49152, zero);
      }
      continue;
    }
    if (mode == 1) {
      const int bx = blockIdx.x & 15, by = blockIdx.x >> 4, rx_n = 4 / tw;
      ux = bx * rx_n + i % rx_n; uy = by * 64 + i / rx_n;
    } else {
      const int u = i * 256 + (int)blockIdx.x;
      ux = u % ux_n; uy = u / ux_n;
    }
    char* base = img + ((size_t)uy * 4 * 4096 + (size_t)ux * tw * 64) * 12 + lane * 12;
    for (int p = 0; p < 4; ++p)
      for (int q = 0; q < tw; ++q) st3(base + (size_t)p * 49152 + q * 768, zero);
    if (mode == 3)
      for (int k = 0; k < fma; ++k) x = x * 1.0000001f + 1e-7f;
  }
  if (x == 12345.678f) sink[0] = x;
}

int main() {
  float* d_buf; float* d_sink;
  CHECK(hipMalloc(&d_buf, (size_t)4096 * 4096 * 12));
  CHECK(hipMalloc(&d_sink, 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  struct Case { int mode, tw, fma; };
  const Case cases[] = {{0, 1, 0}, {1, 1, 0}, {1, 2, 0}, {1, 4, 0}, {2, 1, 0}, {2, 2, 0}, {2, 4, 0},
                        {4, 1, 0}, {5, 1, 0}, {6, 1, 0}, {0, 1, 0}};
  for (int rep = 0; rep < 2; ++rep) {
    for (const Case& c : cases) {
      for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k_pattern, dim3(256), dim3(1024), 0, 0, d_buf, c.mode, c.tw, c.fma, d_sink);
      CHECK(hipEventRecord(e0, 0));
      for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k_pattern, dim3(256), dim3(1024), 0, 0, d_buf, c.mode, c.tw, c.fma, d_sink);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipDeviceSynchronize());
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("mode %d  tiles per unit %d  fma %4d : %6.2f us per launch  (%.2f TB/s)\n", c.mode, c.tw, c.fma, ms * 50.0,
             201.326592 / (ms * 50.0));
    }
  }
  return 0;
}
