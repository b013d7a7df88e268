// Microbenchmark: VALU issue cost per instruction CLASS in shader CYCLES (s_memtime around the
// loop, so DVFS does not enter), at 1 / 2 / 4 / 8 waves per SIMD.  Follow-up to issue_probe.hip,
// which showed (a) VALU instructions with an SGPR operand issuing ~1.5x slower than VGPR-only ones
// once several waves share a SIMD and (b) v_mul/v_add/v_max slower than v_fma.  Each probe is one
// asm block of 8 instructions on 8 independent values; the table prints cycles per instruction per
// SIMD (= cycles per wave-iteration / 8 / waves-per-SIMD... measured as the median over waves of
// (t1 - t0) / iterations, divided by 8 and multiplied by nothing: a wave's own stream; the
// per-SIMD cost is that divided by the number of waves sharing the SIMD).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/issue_probe2.hip -o build/ub/issue_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
constexpr int ITERS = 4096;

#define PROBE(NAME, BODY)                                                                          \
  __global__ void NAME(unsigned long long* out, float a, float b) {                                 \
    float x0 = threadIdx.x * 0.001f + 1.0f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4,     \
          x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;                                                     \
    float one = 1.0f, zero = 0.0f;                                                                  \
    asm volatile("" : "+v"(one), "+v"(zero));                                                       \
    unsigned long long t0, t1;                                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                     \
    for (int it = 0; it < ITERS; ++it) {                                                           \
      asm volatile(BODY                                                                            \
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                   : "v"(a), "v"(b), "s"(a), "s"(b), "v"(one), "v"(zero)                            \
                   : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");                \
    }                                                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                     \
    float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                                 \
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = (t1 - t0) + (s == 12345.f); \
  }

// %8 = v(a) %9 = v(b) %10 = s(a) %11 = s(b) %12 = v(1.0) %13 = v(0.0)
#define R8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define I_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n\t"
#define I_MUL32(i) "v_mul_f32_e32 %" #i ", %" #i ", %8\n\t"
#define I_MUL64(i) "v_mul_f32_e64 %" #i ", %" #i ", %8\n\t"
#define I_MULFMA(i) "v_fma_f32 %" #i ", %" #i ", %8, 0\n\t"
#define I_MULFMAV(i) "v_fma_f32 %" #i ", %" #i ", %8, %13\n\t"
#define I_ADD32(i) "v_add_f32_e32 %" #i ", %" #i ", %8\n\t"
#define I_ADD64(i) "v_add_f32_e64 %" #i ", %" #i ", %8\n\t"
#define I_ADDFMA(i) "v_fma_f32 %" #i ", %" #i ", 1.0, %8\n\t"
#define I_ADDFMAV(i) "v_fma_f32 %" #i ", %" #i ", %12, %8\n\t"
#define I_SUB(i) "v_sub_f32_e32 %" #i ", %" #i ", %8\n\t"
#define I_MAX(i) "v_max_f32_e32 %" #i ", %" #i ", %8\n\t"
#define I_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n\t"
#define I_MOV(i) "v_mov_b32_e32 %" #i ", %8\n\t"
#define I_AND(i) "v_and_b32_e32 %" #i ", %" #i ", %8\n\t"
#define I_XOR(i) "v_xor_b32_e32 %" #i ", %" #i ", %8\n\t"
#define I_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n\t"
#define I_ADDU(i) "v_add_u32_e32 %" #i ", %" #i ", %8\n\t"
#define I_LSHL(i) "v_lshlrev_b32_e32 %" #i ", 1, %" #i "\n\t"
#define I_CNDV(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc\n\t"
#define I_CNDS(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[20:21]\n\t"
#define I_CMPV(i) "v_cmp_lt_f32_e32 vcc, %" #i ", %8\n\t"
#define I_CMPS(i) "v_cmp_lt_f32_e64 s[20:21], %" #i ", %8\n\t"
#define I_CVT(i) "v_cvt_f32_i32_e32 %" #i ", %" #i "\n\t"
#define I_RCP(i) "v_rcp_f32_e32 %" #i ", %" #i "\n\t"
#define I_RSQ(i) "v_rsq_f32_e32 %" #i ", %" #i "\n\t"
#define I_EXP(i) "v_exp_f32_e32 %" #i ", %" #i "\n\t"
#define I_LOG(i) "v_log_f32_e32 %" #i ", %" #i "\n\t"
// SGPR operand forms
#define I_FMA_S1(i) "v_fma_f32 %" #i ", %" #i ", %10, %9\n\t"
#define I_FMA_S2(i) "v_fma_f32 %" #i ", %10, %" #i ", %9\n\t"
#define I_FMA_S3(i) "v_fma_f32 %" #i ", %" #i ", %8, %10\n\t"
#define I_FMAC_S(i) "v_fmac_f32_e32 %" #i ", %10, %9\n\t"
#define I_MUL_S(i) "v_mul_f32_e32 %" #i ", %10, %" #i "\n\t"
#define I_ADD_S(i) "v_add_f32_e32 %" #i ", %10, %" #i "\n\t"
#define I_FMA_LIT(i) "v_fmaak_f32 %" #i ", %" #i ", %8, 0x3f317218\n\t"
#define I_FMA_INL(i) "v_fma_f32 %" #i ", %" #i ", 2.0, %9\n\t"
#define I_FMA_NEG(i) "v_fma_f32 %" #i ", -%" #i ", %8, %9\n\t"
#define I_DPP(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
#define I_RDL(i) "v_readlane_b32 s22, %" #i ", 5\n\t"
#define I_SWAP(i) "v_permlane32_swap_b32 %" #i ", %8\n\t"

PROBE(fma, R8(I_FMA))
PROBE(fmac, R8(I_FMAC))
PROBE(mul_e32, R8(I_MUL32))
PROBE(mul_e64, R8(I_MUL64))
PROBE(mul_as_fma_inl0, R8(I_MULFMA))
PROBE(mul_as_fma_v0, R8(I_MULFMAV))
PROBE(add_e32, R8(I_ADD32))
PROBE(add_e64, R8(I_ADD64))
PROBE(add_as_fma_inl1, R8(I_ADDFMA))
PROBE(add_as_fma_v1, R8(I_ADDFMAV))
PROBE(sub_e32, R8(I_SUB))
PROBE(max_e32, R8(I_MAX))
PROBE(med3, R8(I_MED3))
PROBE(mov, R8(I_MOV))
PROBE(and_b32, R8(I_AND))
PROBE(xor_b32, R8(I_XOR))
PROBE(and_or_b32, R8(I_ANDOR))
PROBE(add_u32, R8(I_ADDU))
PROBE(lshlrev, R8(I_LSHL))
PROBE(cndmask_vcc, R8(I_CNDV))
PROBE(cndmask_sgpr, R8(I_CNDS))
PROBE(cmp_vcc, R8(I_CMPV))
PROBE(cmp_sgpr, R8(I_CMPS))
PROBE(cvt_f32_i32, R8(I_CVT))
PROBE(rcp, R8(I_RCP))
PROBE(rsq, R8(I_RSQ))
PROBE(exp, R8(I_EXP))
PROBE(log, R8(I_LOG))
PROBE(fma_sgpr_src1, R8(I_FMA_S1))
PROBE(fma_sgpr_src0, R8(I_FMA_S2))
PROBE(fma_sgpr_src2, R8(I_FMA_S3))
PROBE(fmac_sgpr, R8(I_FMAC_S))
PROBE(mul_sgpr, R8(I_MUL_S))
PROBE(add_sgpr, R8(I_ADD_S))
PROBE(fmaak_literal, R8(I_FMA_LIT))
PROBE(fma_inline_const, R8(I_FMA_INL))
PROBE(fma_neg_modifier, R8(I_FMA_NEG))
PROBE(dpp_add, R8(I_DPP))
PROBE(readlane, R8(I_RDL))
PROBE(permlane32_swap, R8(I_SWAP))
// mixes: 4 + 4
PROBE(mix_fma4_mul4, I_FMA(0) I_MUL32(1) I_FMA(2) I_MUL32(3) I_FMA(4) I_MUL32(5) I_FMA(6) I_MUL32(7))
PROBE(mix_fma4_sgprfma4, I_FMA(0) I_FMA_S1(1) I_FMA(2) I_FMA_S1(3) I_FMA(4) I_FMA_S1(5) I_FMA(6) I_FMA_S1(7))
PROBE(mix_fma7_rcp1, I_FMA(0) I_FMA(1) I_FMA(2) I_RCP(3) I_FMA(4) I_FMA(5) I_FMA(6) I_FMA(7))
PROBE(mix_fma6_cmp_cnd, I_FMA(0) I_FMA(1) I_CMPV(2) I_FMA(3) I_FMA(4) I_CNDV(5) I_FMA(6) I_FMA(7))

struct Entry { const char* name; void (*fn)(unsigned long long*, float, float); };
#define E(NAME) {#NAME, NAME}

int main(int argc, char** argv) {
  unsigned long long* out;
  const int max_waves = 256 * 8 * 4;
  if (hipMalloc(&out, max_waves * sizeof(unsigned long long)) != hipSuccess) return 1;
  std::vector<unsigned long long> h(max_waves);
  Entry es[] = {E(fma), E(fmac), E(mul_e32), E(mul_e64), E(mul_as_fma_inl0), E(mul_as_fma_v0), E(add_e32), E(add_e64),
                E(add_as_fma_inl1), E(add_as_fma_v1), E(sub_e32), E(max_e32), E(med3), E(mov), E(and_b32), E(xor_b32),
                E(and_or_b32), E(add_u32), E(lshlrev), E(cndmask_vcc), E(cndmask_sgpr), E(cmp_vcc), E(cmp_sgpr),
                E(cvt_f32_i32), E(rcp), E(rsq), E(exp), E(log), E(fma_sgpr_src1), E(fma_sgpr_src0), E(fma_sgpr_src2),
                E(fmac_sgpr), E(mul_sgpr), E(add_sgpr), E(fmaak_literal), E(fma_inline_const), E(fma_neg_modifier),
                E(dpp_add), E(readlane), E(permlane32_swap), E(mix_fma4_mul4), E(mix_fma4_sgprfma4), E(mix_fma7_rcp1),
                E(mix_fma6_cmp_cnd)};
  printf("cycles per instruction per SIMD (median wave's cycles per iteration / 8 / waves per SIMD)\n");
  printf("%-22s %8s %8s %8s %8s\n", "probe", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
  for (auto& e : es) {
    printf("%-22s", e.name);
    fflush(stdout);
    for (int wps = 1; wps <= 8; wps *= 2) {
      const int blocks = 256 * wps;   // 256-thread workgroups: one wave per SIMD each
      for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        if (hipDeviceSynchronize() != hipSuccess) { printf(" launch failed\n"); return 1; }
      }
      const int n = blocks * 4;
      hipMemcpy(h.data(), out, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.begin() + n);
      printf(" %8.2f", (double)h[n / 2] / ITERS / 8.0 / wps);
      fflush(stdout);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
