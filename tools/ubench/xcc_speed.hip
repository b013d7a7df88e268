// xcc_speed.hip -- does every XCD of an MI355X run the same work at the same speed?
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/xcc_speed.hip -o build/ubench/xcc_speed && build/ubench/xcc_speed
//
// One 1024-thread workgroup per CU (256), three phases per launch, timed per workgroup with s_memrealtime (100 MHz,
// the same counter on every XCD) and s_memtime (shader clock):
//   valu   a fixed chain-free FMA loop (no memory)
//   store  every workgroup writes its own contiguous 768 KB (dwordx3 per lane, like k_fwd's rows)
//   load   every workgroup reads its own contiguous 768 KB
// Prints, per hardware XCC_ID, the median duration of each phase and the shader clock seen (d memtime / d memrealtime).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Rec {
  unsigned long long rt[4], clk[4];
  unsigned xcc, pad;
};

__device__ __forceinline__ unsigned long long realtime() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ unsigned long long shadertime() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

struct F3 { float x, y, z; };

__global__ __launch_bounds__(1024) void k_probe(Rec* rec, float* buf, size_t floats_per_wg, int valu_iters, float* sink) {
  const int lane = threadIdx.x;
  Rec r;
  r.xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15;
  r.rt[0] = realtime(); r.clk[0] = shadertime();
  // valu: 8 independent FMA chains
  float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
  const float m = 1.0000001f, c = 1e-7f;
#pragma unroll 1
  for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c;
      a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
    }
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) sink[0] = 1.f;
  __syncthreads();
  r.rt[1] = realtime(); r.clk[1] = shadertime();
  // store: the workgroup's own slab, 12 B per lane and instruction
  F3* dst = reinterpret_cast<F3*>(buf + (size_t)blockIdx.x * floats_per_wg);
  const size_t n3 = floats_per_wg / 3;
  for (size_t i = lane; i < n3; i += 1024) dst[i] = F3{0.f, 0.f, 0.f};
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  r.rt[2] = realtime(); r.clk[2] = shadertime();
  // load: the same slab
  float acc = 0.f;
  const float4* src = reinterpret_cast<const float4*>(buf + (size_t)blockIdx.x * floats_per_wg);
  for (size_t i = lane; i < floats_per_wg / 4; i += 1024) { float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.f) sink[1] = 1.f;
  __syncthreads();
  r.rt[3] = realtime(); r.clk[3] = shadertime();
  if (lane == 0) rec[blockIdx.x] = r;
}

static double median(std::vector<double> v) {
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main() {
  const int n_wg = 256;
  const size_t floats_per_wg = 196608;   // 768 KB
  Rec* d_rec; float* d_buf; float* d_sink;
  CHECK(hipMalloc(&d_rec, n_wg * sizeof(Rec)));
  CHECK(hipMalloc(&d_buf, n_wg * floats_per_wg * sizeof(float)));
  CHECK(hipMalloc(&d_sink, 16));
  std::vector<Rec> rec(n_wg);
  for (int rep = 0; rep < 4; ++rep) {
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k_probe, dim3(n_wg), dim3(1024), 0, 0, d_rec, d_buf, floats_per_wg, 150, d_sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(rec.data(), d_rec, n_wg * sizeof(Rec), hipMemcpyDeviceToHost));
    printf("launch set %d (20 launches, the last one read back)\n", rep);
    for (unsigned x = 0; x < 8; ++x) {
      std::vector<double> tv, ts, tl, ck;
      std::vector<int> blocks;
      for (int b = 0; b < n_wg; ++b) {
        if (rec[b].xcc != x) continue;
        tv.push_back((rec[b].rt[1] - rec[b].rt[0]) / 100.0);
        ts.push_back((rec[b].rt[2] - rec[b].rt[1]) / 100.0);
        tl.push_back((rec[b].rt[3] - rec[b].rt[2]) / 100.0);
        ck.push_back((double)(rec[b].clk[1] - rec[b].clk[0]) / (double)(rec[b].rt[1] - rec[b].rt[0]) * 0.1);
        if (blocks.size() < 3) blocks.push_back(b);
      }
      printf("  XCC %u: %3zu workgroups (blocks %d %d %d ...)  valu %.2f us  store %.2f us  load %.2f us  shader clock %.3f GHz\n", x,
             tv.size(), blocks.size() > 0 ? blocks[0] : -1, blocks.size() > 1 ? blocks[1] : -1, blocks.size() > 2 ? blocks[2] : -1,
             median(tv), median(ts), median(tl), median(ck));
    }
  }
  return 0;
}
