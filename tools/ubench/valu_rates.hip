// Microbenchmark: issue cost of the VALU instruction classes the adjoint uses, on gfx950.
// Eight independent chains per lane, 4 waves per SIMD (the render kernels' occupancy);
// prints cycles per wave-instruction per SIMD (at the measured 2.2-2.4 GHz).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int ITERS = 2048;
#define KERNEL(NAME, ASM)                                                                  \
  __global__ void NAME(float* out, float a, float b) {                                      \
    float x[8];                                                                             \
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i + 1.0f;                     \
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(x[0]), "v"(a) : "vcc");                  \
    for (int it = 0; it < ITERS; ++it) {                                                    \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b) : "vcc", "s20", "s21", "s22", "s23"); \
    }                                                                                       \
    float s = 0;                                                                            \
    for (int i = 0; i < 8; ++i) s += x[i];                                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                         \
  }
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL(k_add, "v_add_f32 %0, %0, %1")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_max, "v_max_f32 %0, %0, %1")
KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
// (no clobber list: the select only READS the mask)
#define KERNEL_RO(NAME, ASM)                                                               \
  __global__ void NAME(float* out, float a, float b) {                                      \
    float x[8];                                                                             \
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i + 1.0f;                     \
    asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 s[20:21], %0, %1" ::"v"(x[0]), "v"(a) : "vcc", "s20", "s21"); \
    for (int it = 0; it < ITERS; ++it) {                                                    \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(a), "v"(b)); \
    }                                                                                       \
    float s = 0;                                                                            \
    for (int i = 0; i < 8; ++i) s += x[i];                                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                         \
  }
KERNEL_RO(k_cndmask_ro, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_RO(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL_RO(k_cndmask_2src, "v_cndmask_b32 %0, %1, %2, vcc")
KERNEL_RO(k_cndmask_e64_vcc, "v_cndmask_b32_e64 %0, %0, %1, vcc")
KERNEL(k_cmp_cnd_sgpr, "v_cmp_lt_f32 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]")
KERNEL(k_cmp_and_cnd, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_gt_f32 s[20:21], %0, %2\n\ts_and_b64 vcc, vcc, s[20:21]\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL(k_cmp_and_cnd_sgpr, "v_cmp_lt_f32 s[22:23], %0, %1\n\tv_cmp_gt_f32 s[20:21], %0, %2\n\ts_and_b64 s[20:21], s[22:23], s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]")
KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL(k_cmp_vcc, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL(k_cmp_sgpr, "v_cmp_lt_f32 s[20:21], %0, %1")
KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_rsq, "v_rsq_f32 %0, %0")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL(k_exp, "v_exp_f32 %0, %0")
KERNEL(k_log, "v_log_f32 %0, %0")
KERNEL(k_and, "v_and_b32 %0, %0, %1")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 5")
KERNEL(k_dpp_add, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_dpp_rowshr, "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf")
KERNEL(k_swap32, "v_permlane32_swap_b32 %0, %1")
KERNEL(k_fma_lit, "v_fmaak_f32 %0, %0, %1, 0x3f317218")

struct Entry { const char* name; void (*fn)(float*, float, float); };
int main() {
  float* out;
  hipMalloc(&out, 256 * 4 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  Entry es[] = {{"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_mov_b32", k_mov},
                {"v_max_f32", k_max}, {"v_med3_f32", k_med3}, {"v_cndmask_b32 (vcc)", k_cndmask}, {"v_cndmask (vcc, read only)", k_cndmask_ro}, {"v_cndmask_e64 (sgpr pair)", k_cndmask_sgpr},
                {"v_cndmask (vcc, 2 other srcs)", k_cndmask_2src}, {"v_cmp -> vcc + v_cndmask (pair)", k_cmp_cnd}, {"v_cndmask_e64 (vcc)", k_cndmask_e64_vcc},
                {"v_cmp -> sgpr + v_cndmask_e64 (pair)", k_cmp_cnd_sgpr}, {"2 v_cmp + s_and + v_cndmask vcc (3 VALU)", k_cmp_and_cnd},
                {"2 v_cmp + s_and + v_cndmask_e64 sgpr (3 VALU)", k_cmp_and_cnd_sgpr},
                {"v_cmp_lt_f32 -> vcc", k_cmp_vcc},
                {"v_cmp_lt_f32 -> sgpr", k_cmp_sgpr}, {"v_cvt_f32_i32", k_cvt_f32_i32}, {"v_cvt_i32_f32", k_cvt_i32_f32},
                {"v_rcp_f32", k_rcp}, {"v_rsq_f32", k_rsq}, {"v_sqrt_f32", k_sqrt}, {"v_exp_f32", k_exp}, {"v_log_f32", k_log},
                {"v_and_b32", k_and}, {"v_lshlrev_b32", k_lshl}, {"v_add_u32", k_add_u32}, {"v_mul_lo_u32", k_mul_lo},
                {"v_readlane_b32", k_readlane}, {"v_add_f32 dpp quad_perm", k_dpp_add}, {"v_add_f32 dpp row_shr", k_dpp_rowshr},
                {"v_permlane32_swap_b32", k_swap32}, {"v_fmaak_f32 (literal)", k_fma_lit}};
  const int blocks = 256 * 4;      // 4 workgroups of 4 waves per CU = 4 waves per SIMD
  for (auto& e : es) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    double inst_per_simd = (double)blocks * 4 * ITERS * 8 / 1024;
    printf("%-28s %.2f ns per wave-instruction per SIMD = %.2f cycles at 2.3 GHz\n", e.name, ms * 1e6 / inst_per_simd,
           ms * 1e6 / inst_per_simd * 2.3);
  }
  return 0;
}
