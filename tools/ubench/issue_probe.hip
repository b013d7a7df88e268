// Microbenchmark: why does k_bwd issue one VALU instruction per ~4.5 cycles per SIMD when plain
// v_fma_f32 streams reach ~2.5?  Probes, at 1 / 2 / 4 waves per SIMD, the patterns its ISA is full of:
//   * a compare and the select that consumes its mask, k independent FMAs apart (is the
//     VALU -> SGPR/VCC -> VALU dependency a stall the other waves of the SIMD can fill?)
//   * FMAs with an SGPR operand (shape parameters live in SGPRs) vs VGPR-only
//   * v_cmpx / EXEC round trips (per-lane branches), v_readlane -> use, transcendental -> use
//   * dependent chains (1, 2, 4 independent chains per wave instead of 8)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int ITERS = 1024;

// One asm block per loop iteration: the order inside is fixed.  x0..x7 are eight independent values.
#define PROBE(NAME, NINST, BODY)                                                                  \
  __global__ void NAME(float* out, float a, float b) {                                             \
    float x0 = threadIdx.x * 0.001f + 1.0f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4,     \
          x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;                                                     \
    for (int it = 0; it < ITERS; ++it) {                                                           \
      asm volatile(BODY                                                                            \
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) \
                   : "v"(a), "v"(b), "s"(a), "s"(b)                                                 \
                   : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");         \
    }                                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;             \
  }                                                                                                \
  static const int NAME##_n = NINST;

#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define FMAS(i) "v_fma_f32 %" #i ", %" #i ", %10, %9\n\t"       // one SGPR operand
#define MULS(i) "v_mul_f32 %" #i ", %10, %" #i "\n\t"           // VOP2, SGPR src0
#define CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n\t"
#define CND(i) "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
#define CMPS(i, s) "v_cmp_lt_f32 " s ", %" #i ", %8\n\t"
#define CNDS(i, s) "v_cndmask_b32_e64 %" #i ", %" #i ", %9, " s "\n\t"

// baseline streams (8 instructions per iteration)
PROBE(p_fma8, 8, FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7))
PROBE(p_fma8_sgpr, 8, FMAS(0) FMAS(1) FMAS(2) FMAS(3) FMAS(4) FMAS(5) FMAS(6) FMAS(7))
PROBE(p_mul8_sgpr, 8, MULS(0) MULS(1) MULS(2) MULS(3) MULS(4) MULS(5) MULS(6) MULS(7))
// dependent chains: the same 8 FMAs on 1, 2, 4 values
PROBE(p_fma_dep1, 8, FMA(0) FMA(0) FMA(0) FMA(0) FMA(0) FMA(0) FMA(0) FMA(0))
PROBE(p_fma_dep2, 8, FMA(0) FMA(1) FMA(0) FMA(1) FMA(0) FMA(1) FMA(0) FMA(1))
PROBE(p_fma_dep4, 8, FMA(0) FMA(1) FMA(2) FMA(3) FMA(0) FMA(1) FMA(2) FMA(3))
// compare -> select, k FMAs between them (8 instructions per iteration: 1 cmp + 6 fma + 1 cnd, order varies)
PROBE(p_cmp_cnd_k0, 8, CMP(0) CND(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6))
PROBE(p_cmp_cnd_k1, 8, CMP(0) FMA(1) CND(0) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6))
PROBE(p_cmp_cnd_k2, 8, CMP(0) FMA(1) FMA(2) CND(0) FMA(3) FMA(4) FMA(5) FMA(6))
PROBE(p_cmp_cnd_k4, 8, CMP(0) FMA(1) FMA(2) FMA(3) FMA(4) CND(0) FMA(5) FMA(6))
PROBE(p_cmp_cnd_k6, 8, CMP(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) CND(0))
// the same with the mask in an SGPR pair
PROBE(p_cmps_cnd_k0, 8, CMPS(0, "s[20:21]") CNDS(0, "s[20:21]") FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6))
PROBE(p_cmps_cnd_k6, 8, CMPS(0, "s[20:21]") FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) CNDS(0, "s[20:21]"))
// compare alone / select alone among FMAs
PROBE(p_cmp_only, 8, CMP(0) FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6))
PROBE(p_cnd_only, 8, CND(0) FMA(7) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6))
// four pairs back to back (what a clip mask looks like) vs software-pipelined across four SGPR pairs
PROBE(p_4pairs_vcc, 8, CMP(0) CND(0) CMP(1) CND(1) CMP(2) CND(2) CMP(3) CND(3))
PROBE(p_4pairs_piped, 8,
      CMPS(0, "s[20:21]") CMPS(1, "s[22:23]") CMPS(2, "s[24:25]") CMPS(3, "s[26:27]")
      CNDS(0, "s[20:21]") CNDS(1, "s[22:23]") CNDS(2, "s[24:25]") CNDS(3, "s[26:27]"))
// arithmetic replacements for a select
PROBE(p_med3_8, 8,
      "v_med3_f32 %0, %0, %8, %9\n\tv_med3_f32 %1, %1, %8, %9\n\tv_med3_f32 %2, %2, %8, %9\n\tv_med3_f32 %3, %3, %8, %9\n\t"
      "v_med3_f32 %4, %4, %8, %9\n\tv_med3_f32 %5, %5, %8, %9\n\tv_med3_f32 %6, %6, %8, %9\n\tv_med3_f32 %7, %7, %8, %9\n\t")
PROBE(p_mul_clamp8, 8,
      "v_mul_f32_e64 %0, |%0|, %8 clamp\n\tv_mul_f32_e64 %1, |%1|, %8 clamp\n\tv_mul_f32_e64 %2, |%2|, %8 clamp\n\t"
      "v_mul_f32_e64 %3, |%3|, %8 clamp\n\tv_mul_f32_e64 %4, |%4|, %8 clamp\n\tv_mul_f32_e64 %5, |%5|, %8 clamp\n\t"
      "v_mul_f32_e64 %6, |%6|, %8 clamp\n\tv_mul_f32_e64 %7, |%7|, %8 clamp\n\t")
PROBE(p_bfi8, 8,
      "v_bfi_b32 %0, %8, %0, %9\n\tv_bfi_b32 %1, %8, %1, %9\n\tv_bfi_b32 %2, %8, %2, %9\n\tv_bfi_b32 %3, %8, %3, %9\n\t"
      "v_bfi_b32 %4, %8, %4, %9\n\tv_bfi_b32 %5, %8, %5, %9\n\tv_bfi_b32 %6, %8, %6, %9\n\tv_bfi_b32 %7, %8, %7, %9\n\t")
// per-lane branch: compare into EXEC, two FMAs under the mask, EXEC restored
PROBE(p_cmpx_2fma, 8,
      "s_mov_b64 s[20:21], exec\n\tv_cmpx_gt_f32 %0, %9\n\t" FMA(1) FMA(2) "s_mov_b64 exec, s[20:21]\n\t" FMA(3) FMA(4) FMA(5)
      FMA(6) FMA(7))
PROBE(p_cmp_saveexec_2fma, 8,
      "v_cmp_gt_f32 vcc, %0, %9\n\ts_and_saveexec_b64 s[20:21], vcc\n\t" FMA(1) FMA(2) "s_mov_b64 exec, s[20:21]\n\t" FMA(3)
      FMA(4) FMA(5) FMA(6) FMA(7))
// v_readlane -> SGPR -> VALU use, adjacent and 6 apart
PROBE(p_readlane_use_k0, 8, "v_readlane_b32 s20, %0, 5\n\tv_mul_f32 %1, s20, %1\n\t" FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7))
PROBE(p_readlane_use_k6, 8, "v_readlane_b32 s20, %0, 5\n\t" FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) "v_mul_f32 %1, s20, %1\n\t")
// transcendental -> use, adjacent and 6 apart
PROBE(p_rcp_use_k0, 8, "v_rcp_f32 %0, %0\n\tv_mul_f32 %0, %0, %8\n\t" FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6))
PROBE(p_rcp_use_k6, 8, "v_rcp_f32 %0, %0\n\t" FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) "v_mul_f32 %0, %0, %8\n\t")
// DPP add and permlane swap among FMAs
PROBE(p_dpp_use_k0, 8, "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" FMA(0) FMA(1) FMA(2) FMA(3) FMA(4)
      FMA(5) FMA(6))
PROBE(p_fma_then_dpp, 8, FMA(0) "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" FMA(1) FMA(2) FMA(3) FMA(4)
      FMA(5) FMA(6))
// LDS read -> use beside FMAs (s_up rows come from LDS)
// scalar ALU beside VALU: does SALU work take VALU issue slots of the same wave?
PROBE(p_fma8_salu8, 8,
      FMA(0) "s_add_u32 s20, s20, 1\n\t" FMA(1) "s_add_u32 s21, s21, 1\n\t" FMA(2) "s_add_u32 s22, s22, 1\n\t" FMA(3)
      "s_add_u32 s23, s23, 1\n\t" FMA(4) "s_add_u32 s24, s24, 1\n\t" FMA(5) "s_add_u32 s25, s25, 1\n\t" FMA(6)
      "s_add_u32 s26, s26, 1\n\t" FMA(7) "s_add_u32 s27, s27, 1\n\t")

struct Entry { const char* name; void (*fn)(float*, float, float); int n; };
#define E(NAME) {#NAME, NAME, NAME##_n}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  Entry es[] = {E(p_fma8), E(p_fma8_sgpr), E(p_mul8_sgpr), E(p_fma_dep1), E(p_fma_dep2), E(p_fma_dep4),
                E(p_cmp_cnd_k0), E(p_cmp_cnd_k1), E(p_cmp_cnd_k2), E(p_cmp_cnd_k4), E(p_cmp_cnd_k6),
                E(p_cmps_cnd_k0), E(p_cmps_cnd_k6), E(p_cmp_only), E(p_cnd_only), E(p_4pairs_vcc), E(p_4pairs_piped),
                E(p_med3_8), E(p_mul_clamp8), E(p_bfi8), E(p_cmpx_2fma), E(p_cmp_saveexec_2fma),
                E(p_readlane_use_k0), E(p_readlane_use_k6), E(p_rcp_use_k0), E(p_rcp_use_k6), E(p_dpp_use_k0),
                E(p_fma_then_dpp), E(p_fma8_salu8)};
  // clock: p_fma8 at 8 waves/SIMD is the reference stream
  printf("%-24s %10s %10s %10s %10s   (ns per iteration of 8 VALU instructions, per SIMD)\n", "probe", "1 w/SIMD", "2 w/SIMD",
         "4 w/SIMD", "8 w/SIMD");
  for (auto& e : es) {
    printf("%-24s", e.name);
    fflush(stdout);
    for (int wps = 1; wps <= 8; wps *= 2) {
      const int blocks = 256 * wps;   // 256-thread workgroups: one wave per SIMD each
      float ms = 0, best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      const double iters_per_simd = (double)blocks * 4 * ITERS / 1024;
      printf(" %10.2f", best * 1e6 / iters_per_simd);
      fflush(stdout);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
