// rrt_kernels.hip -- gfx950 (MI355X, CDNA4) render kernels + the C ABI of rrt.h.
//
// Kernels of one forward/backward pair (DESIGN.md section 5 has the full story
// and the measurements behind every choice):
//
//   k_prep      workgroup = a 4 x 64 block of wave tiles (a tile is 64 x P
//               pixels) of one scene.  Its threads pack the scene's shapes (fold
//               the camera into the transform, per-shape constants, camera-local
//               bounding ball; the scene's first workgroup stores the 128-B
//               records and zeroes the gradient-accumulator rows), then run a
//               two-level cone-vs-ball cull -> a candidate bit mask per tile;
//               packed tile entries go to the workgroup's own segment of a "has
//               candidates" list (ordered by cost class: 3+ candidates, 2, 1) and
//               of a "none" list (no atomics, nothing pre-zeroed).
//   k_fwd<P>    persistent, one 1024-thread workgroup per CU; one WAVE per tile,
//               one lane per pixel column, P rows per lane.  Prologue: a
//               workgroup scan of the segment sizes (DPP) and a binary search
//               per entry deal the workgroup an equal share of every cost class
//               into an LDS work queue, expensive first, and its share of the
//               "none" tiles into a second one (the head of its OWN segments is
//               zero-filled while that is going on).  A wave pops a tile to
//               render and, with it, its part of the zero-fill.  The packed
//               scene is staged in LDS once and a shape's record reaches SGPRs
//               by one ds_read_b32 + v_readlane (scalar loads through the
//               constant address space when the scene is too big); rays are
//               generated in registers; winners are shaded shape by shape; each
//               lane stores its RGB triple with one buffer_store_dwordx3 (a wave
//               row is 768 contiguous bytes), written through (sc1): no dirty
//               lines are left for the kernel's end to write back.  The tiles
//               that are only zero-filled are dealt 6:5 to workgroups of even:odd
//               blockIdx (the odd XCDs of an MI355X store ~15 % slower).  The
//               dealt queue batch is kept in the workspace for the backward.
//   k_bwd<P>    persistent, one 1024-thread workgroup per CU with the same LDS
//               work queue (read back from the forward's batch when there was
//               one); tiles without candidates never touch the upstream
//               gradient; the next tile's gradient rows are requested before the
//               current tile's arithmetic.  For every candidate shape that won
//               somewhere in the wave, the lanes it won run the analytic adjoint
//               with the shape's parameters in SGPRs and add into 22 per-lane
//               accumulators; these are summed across the wave by a transposing
//               butterfly (v_permlane32_swap, v_permlane16_swap, DPP), added to
//               workgroup-level LDS accumulators, and each workgroup adds a
//               touched shape row to HBM once (global_atomic_add_f32) -- to one of
//               four copies of the rows: same-line float atomics serialise.
//               Small images with several samples per pixel (fewer than 4 tiles
//               per wave): every queue entry is handed out 2 or 4 times and the
//               waves share the tile's samples (k_bwd<..., SP = 1>).
//   k_finalize  sum the copies, chain the accumulator rows back to w2o /
//               material / light and re-zero them.
//
// No MFMA: the path is per-ray quadratic solves, not a contraction.
// Environment variables RRT_TUNE_* are tuning knobs for experiments and tests (tile
// rows, persistent grid sizes, LDS staging on/off, queue batch size, saved queues
// on/off, sample split); defaults are what was measured best.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "rrt.h"
#include "rrt_device.h"

namespace {

using namespace rrt;

constexpr int TILE_W = 64;      // one lane per pixel column: a wave row is 768 contiguous bytes
constexpr int WAVES = 16;       // waves per render workgroup: ONE 1024-thread workgroup fills a CU at the
constexpr int BLOCK = WAVES * 64;   // kernels' 4 waves/SIMD, so its LDS work queue balances the whole CU
// k_prep: wave tiles per workgroup (= threads) and list slots per segment
constexpr int BIN_RX = 4, BIN_RY = 64, SEG = BIN_RX * BIN_RY;
constexpr int MAX_WORDS = RRT_MAX_SHAPES / 64;

// Wave-uniform read-only data goes through the constant address space so the
// backend selects s_load_* (scalar cache, SGPR results).
typedef const __attribute__((address_space(4))) float* cfptr;
typedef const __attribute__((address_space(4))) unsigned long long* cuptr;
__device__ __forceinline__ cfptr as_const(const float* p) { return (cfptr)p; }
__device__ __forceinline__ cuptr as_const(const unsigned long long* p) { return (cuptr)p; }
typedef const __attribute__((address_space(4))) unsigned* cu32ptr;
__device__ __forceinline__ cu32ptr as_const_u32(const unsigned* p) { return (cu32ptr)p; }

struct F3 {
  float x, y, z;
};

// Diagnostic build only (-DRRT_STAMPS, tools/stamp_bwd.py): s_memtime stamps of
// k_bwd's phases, written to a buffer nothing else reads.  Never in the shipped
// library (MI355X_MICROARCH.md: stamps go to their own buffer; read shares, not lengths).
#ifdef RRT_STAMPS
__device__ unsigned long long* g_stamps = nullptr;   // [n_tiles][8]
#define RRT_STAMP(slot)                                                                     \
  do {                                                                                      \
    unsigned long long t_;                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    if (g_stamps && lane == 0) g_stamps[tile * 16 + (slot)] = t_;                           \
  } while (0)
#define RRT_STAMP_RT(slot)                                                                  \
  do {                                                                                      \
    unsigned long long t_;                                                                  \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
    if (g_stamps && lane == 0) g_stamps[tile * 16 + (slot)] = t_;                           \
  } while (0)
#define RRT_STAMP_VAL(slot, value)                                                          \
  do {                                                                                      \
    if (g_stamps && lane == 0) g_stamps[tile * 16 + (slot)] = (unsigned long long)(value);  \
  } while (0)
// k_fwd: per-wave real-time stamps / counters in row `gw` (tools/stamp_fwd.py)
#define RRT_STAMP_FWD(slot, value_or_rt)                                                    \
  do {                                                                                      \
    unsigned long long t_ = (value_or_rt);                                                  \
    if (t_ == ~0ull) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    if (g_stamps && lane == 0) g_stamps[(size_t)gw * 8 + (slot)] = t_;                      \
  } while (0)
#else
#define RRT_STAMP(slot) do {} while (0)
#define RRT_STAMP_RT(slot) do {} while (0)
#define RRT_STAMP_VAL(slot, value) do {} while (0)
#define RRT_STAMP_FWD(slot, value_or_rt) do {} while (0)
#endif

struct KArgs {
  ViewK v;
  int S, B, nwords;
  int P;                          // rows per wave tile
  int tiles_x, tiles_y;           // wave tiles per scene
  long long n_tiles;              // B * tiles_y * tiles_x
  int n_waves;                    // waves of a persistent launch (render kernels)
  const float* ps;                // [B][S][PS_STRIDE]
  const float* hdr;               // [B][HDR_STRIDE]
  const unsigned long long* mask; // [n_tiles][nwords]
  const unsigned long long* heavy; // [n_tiles] packed entries of tiles with >= 1 candidate (first counts[0])
  const unsigned long long* empty; // [n_tiles] packed entries of tiles with none           (first counts[1])
  int lds_scene;                  // 1: all B*S packed records fit the kernels' LDS staging area (and the LDS accumulators);
                                  // 2: too many for that, but one SCENE's do: a workgroup takes whole list segments
                                  //    (a segment lies in one scene), stages that scene and accumulates per segment;
                                  // 0: neither (a scene of more than LDS_RECS shapes): scalar loads, global atomics
  int bins_per_scene;             // list segments per scene
  const unsigned* blk_counts;     // [n_bins][2] per prep-block: #tiles with 3+ candidates | 2 candidates << 16,
                                  //             1 candidate | none << 16
  int n_bins;                     // prep workgroups = list segments of SEG slots
  unsigned long long* wgq;        // [batches * workgroups][Q_CAP] the queue batches the last render kernel dealt itself,
  unsigned* wgq_cnt;              // [batches * workgroups]        kept so that a backward on the same setup reads its
  int use_saved;                  // share back (1) instead of redoing scans, searches and the entry gather
  int chunk;                      // list segments per queue batch: at most one per thread, and few enough that a
                                  // workgroup's share of their tiles fits its queues (Q_CAP)
  int zw_even, zw_odd;            // k_fwd: shares of the zero-filled tiles, workgroups with even : odd blockIdx (XcdDeal)
  int ztail;                      // k_fwd: per cent of a workgroup's zero-filled tiles kept back for its waves' idle ends
  const float* jitter;            // explicit sampleDist or nullptr
  float* image;                   // fwd: [B][rows][ncols][3]
  const float* grad;              // bwd: same layout
  float* acc;                     // bwd: [acc_repl][B][S][ACC_STRIDE]
  int split_log2;                 // k_bwd<SP = 1>: a queue entry is pulled 1 << split_log2 times, each wave taking an equal part of
                                  // the tile's antialiasing samples (small images: a wave's one or two tiles are the critical path)
  int acc_repl;                   // copies of the accumulator rows: workgroup b of k_bwd adds to copy b mod acc_repl and
                                  // k_finalize sums the copies.  Same-line float atomics serialise at ~13 ns each: with one
                                  // copy all 256 workgroups queue on every shape's line -- 3.5 us at the end of k_bwd
};
__device__ __forceinline__ float* acc_copy(const KArgs& a) {      // this workgroup's copy of the accumulator rows
  return a.acc + (size_t)(blockIdx.x % (unsigned)a.acc_repl) * ((size_t)a.B * a.S * ACC_STRIDE);
}

// The render kernels carry ~170 scalar values over their tile loops and spill them into vector lanes; every use is then a
// v_readlane, a VALU instruction of the slow class.  What a tile needs of the kernel arguments only at its start or end is
// therefore read from the kernarg segment where it is used (a scalar load, off the VALU): the `asm` keeps the compiler from
// hoisting the loads to the kernel's top.  Every kernel here takes the KArgs struct first (offset 0 of the segment).
typedef const __attribute__((address_space(4))) KArgs* ckargs;
__device__ __forceinline__ ckargs kargs_here() {
  ckargs p = (ckargs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

struct Tile {
  int b, a0, b0, nrows, ncols;    // scene, first row (window-relative), first column, extent
};

// Packed tile-list entry: tx[0:16) ty[16:36) b[36:52) first candidate k0[52:62) multi[62]
// (multi = more than one candidate shape: the mask words must be consulted).
constexpr int ENT_TX_BITS = 16, ENT_TY_BITS = 20, ENT_B_BITS = 16;
__device__ __forceinline__ unsigned long long encode_entry(int tx, int ty, int b, int k0, bool multi) {
  return (unsigned long long)tx | ((unsigned long long)ty << 16) | ((unsigned long long)b << 36) |
         ((unsigned long long)k0 << 52) | ((unsigned long long)multi << 62);
}
__device__ __forceinline__ Tile decode_entry(const KArgs& a, unsigned long long e, int& k0, bool& multi,
                                             long long& tile) {
  const int tx = (int)(e & 0xffffu), ty = (int)((e >> 16) & 0xfffffu);
  Tile t;
  t.b = (int)((e >> 36) & 0xffffu);
  k0 = (int)((e >> 52) & 0x3ffu);
  multi = ((e >> 62) & 1u) != 0;
  ckargs ka = kargs_here();
  t.a0 = ty * ka->P;
  t.b0 = tx * TILE_W;
  t.nrows = min(ka->P, ka->v.rows - t.a0);
  t.ncols = min(TILE_W, ka->v.ncols - t.b0);
  tile = ((long long)t.b * ka->tiles_y + ty) * ka->tiles_x + tx;
  return t;
}

__device__ __forceinline__ Tile decode_tile(const KArgs& a, long long tile) {
  Tile t;
  int tx = (int)(tile % a.tiles_x);
  long long rest = tile / a.tiles_x;
  int ty = (int)(rest % a.tiles_y);
  t.b = (int)(rest / a.tiles_y);
  t.a0 = ty * a.P;
  t.b0 = tx * TILE_W;
  t.nrows = min(a.P, a.v.rows - t.a0);
  t.ncols = min(TILE_W, a.v.ncols - t.b0);
  return t;
}

// ---------------------------------------------------------------- helpers --
constexpr int LDS_RECS = 128;          // packed shape records a workgroup stages in LDS (16 KB)

__device__ __forceinline__ float lane_bcast(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ unsigned long long lane_bcast64(unsigned long long v, int src) {
  unsigned lo = __builtin_amdgcn_readlane((unsigned)v, src);
  unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}

// Index of shape k of scene b in the LDS copy of the scene(s): the whole batch (lds_scene = 1) or only the
// scene of the list segment the workgroup is working on (lds_scene = 2).
__device__ __forceinline__ size_t lds_rec(const KArgs& a, int b, int k) {
  ckargs ka = kargs_here();
  return (size_t)(ka->lds_scene == 2 ? 0 : b) * ka->S + k;
}

// One shape's packed record as wave-uniform scalars (SGPRs).
struct ShapeP {
  float A[9], o[3], colI[3], ks, kd, ka, sh;
  float cmax, kssh;               // max colI, ks * sh (the FAST_* code paths)
  int kind, powf;
};

// From the LDS copy of the scene: ONE conflict-free ds_read_b32 (lane l reads
// word l of the 32-word record) + v_readlane broadcasts -- ~100 cycles instead
// of a scalar-cache / L2 round trip per visit.  Falls back to scalar loads
// when the scene does not fit the staging area.
__device__ __forceinline__ void load_shape(const KArgs& a, const float* lds, int b, int k, int lane, ShapeP& q) {
  if (a.lds_scene) {
    const float v = lds[lds_rec(a, b, k) * PS_STRIDE + (lane & 31)];
#pragma unroll
    for (int n = 0; n < 9; ++n) q.A[n] = lane_bcast(v, PS_A + n);
#pragma unroll
    for (int n = 0; n < 3; ++n) { q.o[n] = lane_bcast(v, PS_O + n); q.colI[n] = lane_bcast(v, PS_COLI + n); }
    q.ks = lane_bcast(v, PS_KS); q.kd = lane_bcast(v, PS_KD); q.ka = lane_bcast(v, PS_KA); q.sh = lane_bcast(v, PS_SH);
    q.kind = ps_load_int(lane_bcast(v, PS_KIND));
    q.powf = ps_load_int(lane_bcast(v, PS_POWF));
    q.cmax = lane_bcast(v, PS_CMAX); q.kssh = lane_bcast(v, PS_KSSH);
  } else {
    cfptr r = as_const(a.ps) + ((size_t)b * a.S + k) * PS_STRIDE;
#pragma unroll
    for (int n = 0; n < 9; ++n) q.A[n] = r[PS_A + n];
#pragma unroll
    for (int n = 0; n < 3; ++n) { q.o[n] = r[PS_O + n]; q.colI[n] = r[PS_COLI + n]; }
    q.ks = r[PS_KS]; q.kd = r[PS_KD]; q.ka = r[PS_KA]; q.sh = r[PS_SH];
    q.kind = ps_load_int(r[PS_KIND]);
    q.powf = ps_load_int(r[PS_POWF]);
    q.cmax = r[PS_CMAX]; q.kssh = r[PS_KSSH];
  }
}

// The same in two steps, for callers that decide on kind / flags first and then want only a part:
// `word` is the record as one VGPR (lane l holds word l); the pieces are v_readlane broadcasts of it.
__device__ __forceinline__ float shape_word(const KArgs& a, const float* lds, int b, int k, int lane) {
  ckargs ka = kargs_here();
  return ka->lds_scene ? lds[lds_rec(a, b, k) * PS_STRIDE + (lane & 31)]
                       : ka->ps[((size_t)b * ka->S + k) * PS_STRIDE + (lane & 31)];
}
__device__ __forceinline__ void shape_head(float word, ShapeP& q) {
  q.kind = ps_load_int(lane_bcast(word, PS_KIND));
  q.powf = ps_load_int(lane_bcast(word, PS_POWF));
}
__device__ __forceinline__ void shape_material(float word, ShapeP& q) {
#pragma unroll
  for (int n = 0; n < 3; ++n) q.colI[n] = lane_bcast(word, PS_COLI + n);
  q.ks = lane_bcast(word, PS_KS); q.kd = lane_bcast(word, PS_KD); q.ka = lane_bcast(word, PS_KA); q.sh = lane_bcast(word, PS_SH);
  q.cmax = lane_bcast(word, PS_CMAX); q.kssh = lane_bcast(word, PS_KSSH);
}
__device__ __forceinline__ void shape_transform(float word, ShapeP& q) {
#pragma unroll
  for (int n = 0; n < 9; ++n) q.A[n] = lane_bcast(word, PS_A + n);
#pragma unroll
  for (int n = 0; n < 3; ++n) q.o[n] = lane_bcast(word, PS_O + n);
}

// The transform of a shape as per-lane VGPR copies (every lane reads the same LDS words: three
// broadcast ds_read_b128, A' and o' are the first 12 words of the record).
__device__ __forceinline__ void load_shape_vgpr(const KArgs& a, const float* lds, int b, int k, float Av[9], float ov[3]) {
  // (from global memory when the scene is not staged: still every lane the same address, one request)
  ckargs ka = kargs_here();
  const float4* r = reinterpret_cast<const float4*>(ka->lds_scene ? lds + lds_rec(a, b, k) * PS_STRIDE
                                                                  : ka->ps + ((size_t)b * ka->S + k) * PS_STRIDE);
  const float4 x0 = r[0], x1 = r[1], x2 = r[2];
  Av[0] = x0.x; Av[1] = x0.y; Av[2] = x0.z; Av[3] = x0.w; Av[4] = x1.x; Av[5] = x1.y; Av[6] = x1.z; Av[7] = x1.w;
  Av[8] = x2.x; ov[0] = x2.y; ov[1] = x2.z; ov[2] = x2.w;
#pragma unroll
  for (int n = 0; n < 9; ++n) asm volatile("" : "+v"(Av[n]));
#pragma unroll
  for (int n = 0; n < 3; ++n) asm volatile("" : "+v"(ov[n]));
}

// Per-scene shading constants (wave-uniform): normed light direction, look_at, their dot product.
struct SceneP {
  float L[3], look[3], Llook;
};
// (`ka`: where the kernel arguments are read from.  The default is evaluated in the caller, i.e. inside a kernel;
// bwd_generic_shape -- a real function, in which the kernarg-segment builtin yields NULL -- passes the pointer its kernel
// gave it)
__device__ __forceinline__ void load_scene(const KArgs& a, int b, SceneP& sc, ckargs ka = kargs_here()) {
  cfptr hdr = as_const(ka->hdr) + (size_t)b * HDR_STRIDE;
#pragma unroll
  for (int n = 0; n < 3; ++n) { sc.L[n] = hdr[HDR_LHAT + n]; sc.look[n] = hdr[HDR_LOOK + n]; }
  sc.Llook = hdr[HDR_LLOOK];
}

__device__ __forceinline__ void stage_scene(const KArgs& a, float* lds) {
  if (a.lds_scene == 1) {
    const int n = a.B * a.S * PS_STRIDE;
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = a.ps[i];
  }
  __syncthreads();       // (once, before the tile loop)
}
// lds_scene = 2: the records of ONE scene (the scene of the list segment the workgroup turns to)
__device__ __forceinline__ void stage_one_scene(const KArgs& a, float* lds, int b) {
  const int n = a.S * PS_STRIDE;
  const float* src = a.ps + (size_t)b * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = src[i];
}

// EXTENSION (RRT_FLAG_SHADOWS, default off; rrt_device.h: shadowed_by): is the point this lane's ray hits
// (camera-local direction d, distance t, on shape `self` -- may differ per lane) in the shadow of another
// Sphere of scene b?  A shadow ray can leave the tile's candidate set, so all S shapes are visited; the
// wave stops as soon as none of its lanes is both concerned and still lit -- the wave ballot is the
// any-hit test.  Must be called by all lanes (`active` says which ones are concerned).
__device__ __forceinline__ bool wave_in_shadow(const KArgs& a, const float* lds, int b, int self, const float d[3],
                                               float t, bool active, int lane) {
  bool sh = false;
  for (int k2 = 0; k2 < a.S; ++k2) {
    if (!__ballot(active && !sh)) break;
    const float word = shape_word(a, lds, b, k2, lane);
    if (ps_load_int(lane_bcast(word, PS_KIND)) != KIND_SPHERE) continue;     // only Spheres cast shadows
    float A2[9], o2[3], ell[3];
#pragma unroll
    for (int n = 0; n < 9; ++n) A2[n] = lane_bcast(word, PS_A + n);
#pragma unroll
    for (int n = 0; n < 3; ++n) { o2[n] = lane_bcast(word, PS_O + n); ell[n] = lane_bcast(word, PS_ELL + n); }
    const bool s2 = shadowed_by<float>(A2, o2, ell, lane_bcast(word, PS_ELL2), d, t);
    sh |= active && k2 != self && s2;
  }
  return sh;
}

// Ray generation, split so that what depends only on the lane stays out of the row loops
// (rrt_device.h: ray_dir_parts).  LaneRay: the lane's own grid coordinate cl (u under the root
// camera's transpose, w otherwise), q = cl^2 + 1 and its clamped image column.
struct LaneRay {
  float cl, q;
  int col;
};
template <int TR>
__device__ __forceinline__ LaneRay lane_ray(const KArgs& a, const Tile& t, int lane) {
  LaneRay lr;
  lr.col = t.b0 + min(lane, t.ncols - 1);
  lr.cl = TR ? grid_u<float>(a.v, lr.col) : grid_w<float>(a.v, lr.col);
  lr.q = lr.cl * lr.cl + 1.0f;
  return lr;
}
// sample s of the pixel in window row `row` (JC: jitter at the stratum centre -- jc = sampleDist / dims,
// the same for every pixel, is passed in; otherwise hash or explicit jitter per ray; TR = view.transpose)
template <int JC, int TR>
__device__ __forceinline__ void row_ray(const KArgs& a, const LaneRay& lr, int row, int s, float jcx, float jcy,
                                        float d[3]) {
  const ViewK& v = a.v;
  const int ra = row + v.row0;
  const float cr = TR ? grid_w<float>(v, ra) : grid_u<float>(v, ra);
  float jx = jcx, jy = jcy;
  if (!JC) {
    int i, j;
    pixel_to_ray(v, ra, lr.col, i, j);
    float sdx, sdy;
    sample_dist<float>(v, a.jitter, i, j, s, sdx, sdy);
    jx = sdx * v.inv_xd;
    jy = sdy * v.inv_yd;
  }
  ray_dir_parts<float>(TR ? lr.cl : cr, TR ? cr : lr.cl, lr.q, cr, jx, jy, d);
}
__device__ __forceinline__ void centre_jitter(const ViewK& v, int s, float& jcx, float& jcy) {
  const float sd = (0.5f + (float)s) * v.inv_A;      // sample_dist(JITTER_CENTER)
  jcx = sd * v.inv_xd;
  jcy = sd * v.inv_yd;
}

template <int P, int JC, int TR>
__device__ __forceinline__ void gen_rays(const KArgs& a, const Tile& t, const LaneRay& lr, int s, float d[P][3]) {
  float jcx, jcy;
  centre_jitter(a.v, s, jcx, jcy);
#pragma unroll
  for (int p = 0; p < P; ++p) row_ray<JC, TR>(a, lr, t.a0 + min(p, t.nrows - 1), s, jcx, jcy, d[p]);
}

// One candidate against this lane's P rays (scene.py:38-47: strict '<', the
// caller visits candidates in list order so the first wins ties).
template <int P, bool WANT_NU>
__device__ __forceinline__ void trace_shape_p(int kind, const float* A, const float* o, int k, const float d[P][3],
                                              int best[P], float tmin[P], float nu[P][3]) {
  if (kind == KIND_SPHERE) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float r[3];
      xform_dir<float>(A, d[p], r);
      SphereHit<float> sp;
      bool hit = hit_sphere<float>(r, o, sp);
      if (hit && sp.t < tmin[p]) {
        tmin[p] = sp.t; best[p] = k;
        if (WANT_NU) sphere_point<float>(sp, r, nu[p]);
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float r[3], t;
      xform_dir<float>(A, d[p], r);
      bool hit = hit_square<float>(r, o, t);
      if (hit && t < tmin[p]) {
        tmin[p] = t; best[p] = k;
        if (WANT_NU) { nu[p][0] = 0.f; nu[p][1] = 0.f; nu[p][2] = o[2] > 0.f ? 1.f : -1.f; }
      }
    }
  }
}
template <int P, bool WANT_NU>
__device__ __forceinline__ void trace_shape(const ShapeP& q, int k, const float d[P][3], int best[P], float tmin[P],
                                            float nu[P][3]) {
  trace_shape_p<P, WANT_NU>(q.kind, q.A, q.o, k, d, best, tmin, nu);
}

template <int NW>
__device__ __forceinline__ int global_wave(int& lane) {
  lane = threadIdx.x & 63;
  // provably wave-uniform => SGPR addressing downstream
  return (int)blockIdx.x * NW + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
}

// ------------------------------------------------------------------- prep --
// ONE launch prepares everything the render kernels read from the workspace.
// A workgroup owns a BIN_RX x BIN_RY block of wave tiles of one scene (about
// square in pixels: 256 x 64P):
//   1. every thread packs one shape of the scene (camera folded into the
//      transform, per-shape constants, camera-local bounding ball); the block
//      that owns the scene's first tiles also writes the 128-B records, the
//      scene header and zeroed accumulator rows to the workspace;
//   2. level-1 cull: the same thread tests its shape against the REGION's ray
//      cone; survivors are ballot-compacted into LDS in list order;
//   3. level-2 cull: every thread tests its own TILE against the survivors ->
//      64-bit candidate masks;
//   4. the tile entries (column, row, scene, first candidate, "several" bit)
//      are compacted into the block's own 256-slot segment of the "has
//      candidates" and the "none" list; the block's two counts go to blk_counts.
// No global atomics and nothing that needs pre-zeroing: consumers walk the
// segments (slot s of block s>>8 is valid iff (s & 255) < count).  A shape is
// dropped only when no line through the camera origin inside the tile's
// direction cone can touch its bounding ball.
struct PrepArgs {
  const float* w2o; long long w2o_bs;
  const float* mat; long long mat_bs;
  const float* light;
  const uint8_t* kind;
  const float* cam; long long cam_bs;
  float* ps; float* hdr; float* acc;
  unsigned long long* mask; unsigned long long* heavy; unsigned long long* empty;
  unsigned* blk_counts;
};

__global__ __launch_bounds__(256) void k_prep(KArgs a, PrepArgs pa) {
  __shared__ float4 s_ball[RRT_MAX_SHAPES];     // surviving shapes' bounding balls, list order
  __shared__ unsigned short s_id[RRT_MAX_SHAPES];
  __shared__ int s_wave_n[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bx_n = (a.tiles_x + BIN_RX - 1) / BIN_RX, by_n = (a.tiles_y + BIN_RY - 1) / BIN_RY;
  int blk = blockIdx.x;
  const int bx = blk % bx_n; blk /= bx_n;
  const int by = blk % by_n;
  const int b = blk / by_n;
  const bool owner = bx == 0 && by == 0;        // writes scene b's records
  const int tx = bx * BIN_RX + (tid & (BIN_RX - 1)), ty = by * BIN_RY + (tid / BIN_RX);
  const bool valid = tx < a.tiles_x && ty < a.tiles_y;
  const long long tile = ((long long)b * a.tiles_y + ty) * a.tiles_x + tx;
  const bool nocull = (a.v.flags & RRT_FLAG_NO_CULL) != 0;
  const float* cam = pa.cam + b * pa.cam_bs;

  Cone<float> region;
  {
    int ra0 = by * BIN_RY * a.P, ra1 = min((by + 1) * BIN_RY * a.P, a.v.rows) - 1;
    int rb0 = bx * BIN_RX * TILE_W, rb1 = min((bx + 1) * BIN_RX * TILE_W, a.v.ncols) - 1;
    int i0, j0, i1, j1;
    pixel_to_ray(a.v, ra0 + a.v.row0, rb0, i0, j0);
    pixel_to_ray(a.v, ra1 + a.v.row0, rb1, i1, j1);
    region = tile_cone<float>(a.v, i0, i1, j0, j1);
  }
  // (the scene's header: by a thread of the LAST wave -- the first waves pack shapes, and the owner block, one per CU like
  // every other, is the one the kernel waits for)
  if (owner && tid == 192) pack_header<float, float>(cam, pa.light, pa.hdr + (size_t)b * HDR_STRIDE);
  {
    // the scene's gradient-accumulator rows, every copy: a slice per block of the scene (k_finalize leaves them zeroed
    // too; this covers a fresh workspace)
    const int per = a.S * (ACC_STRIDE / 4), total = per * a.acc_repl;
    const int nblk = bx_n * by_n, me = by * bx_n + bx, slice = (total + nblk - 1) / nblk;
    for (int i = me * slice + tid; i < min(total, (me + 1) * slice); i += blockDim.x) {
      const int r = i / per, j = i - r * per;
      reinterpret_cast<float4*>(pa.acc + ((size_t)r * a.B + b) * a.S * ACC_STRIDE)[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }

  int n = 0;                                     // survivors so far (uniform)
  for (int k0 = 0; k0 < a.S; k0 += 256) {
    const int k = k0 + tid;
    float4 ball = make_float4(0.f, 0.f, 0.f, 0.f);
    bool c = false;
    if (k < a.S) {
      float rec[PS_STRIDE];
      pack_shape<float, float>(pa.w2o + b * pa.w2o_bs + 12 * k, cam, pa.mat + b * pa.mat_bs + 7 * k, pa.light,
                               pa.kind ? (int)pa.kind[k] : 0, rec);
      ball = make_float4(rec[PS_CULL_C], rec[PS_CULL_C + 1], rec[PS_CULL_C + 2], rec[PS_CULL_R]);
      if (owner) {
        float4* dst = reinterpret_cast<float4*>(pa.ps + ((size_t)b * a.S + k) * PS_STRIDE);
#pragma unroll
        for (int q = 0; q < PS_STRIDE / 4; ++q) dst[q] = make_float4(rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]);
      }
      float cc[3] = {ball.x, ball.y, ball.z};
      c = nocull || cone_may_hit<float>(region, cc, ball.w);
    }
    unsigned long long bal = __ballot(c);
    if (lane == 0) s_wave_n[wave] = __builtin_popcountll(bal);
    __syncthreads();
    int off = n;
    for (int w = 0; w < wave; ++w) off += s_wave_n[w];
    if (c) {
      int slot = off + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
      s_ball[slot] = ball;
      s_id[slot] = (unsigned short)k;
    }
    n += s_wave_n[0] + s_wave_n[1] + s_wave_n[2] + s_wave_n[3];
    __syncthreads();
  }

  // level 2: this thread's tile vs the survivors (ascending shape id)
  bool any = false;
  int k_first = 0, n_cand = 0;
  if (valid) {
    Tile t;
    t.b = b; t.a0 = ty * a.P; t.b0 = tx * TILE_W;
    t.nrows = min(a.P, a.v.rows - t.a0); t.ncols = min(TILE_W, a.v.ncols - t.b0);
    int i0, j0, i1, j1;
    pixel_to_ray(a.v, t.a0 + a.v.row0, t.b0, i0, j0);
    pixel_to_ray(a.v, t.a0 + t.nrows - 1 + a.v.row0, t.b0 + t.ncols - 1, i1, j1);
    Cone<float> cone = tile_cone<float>(a.v, i0, i1, j0, j1);
    Frustum<float> fr = tile_frustum<float>(a.v, i0, i1, j0, j1);
    unsigned long long m = 0;
    int cur = 0;                                 // mask word being assembled
    for (int q = 0; q < n; ++q) {
      const int k = s_id[q];
      const int w = k >> 6;
      while (cur < w) { pa.mask[tile * a.nwords + cur] = m; any |= (m != 0); m = 0; ++cur; }
      bool c = true;
      if (!nocull) {
        float4 ball = s_ball[q];
        float cc[3] = {ball.x, ball.y, ball.z};
        c = cone_may_hit<float>(cone, cc, ball.w) && frustum_may_hit<float>(fr, cc, ball.w);
      }
      m |= (unsigned long long)c << (k & 63);
      if (c) { if (!n_cand) k_first = k; ++n_cand; }
    }
    // (the last -- for up to 64 shapes the only -- word; with one word, only tiles with SEVERAL candidates are ever looked
    // up: an entry with a single candidate carries it, and 95 % of the tiles' 8-byte scattered stores stay home)
    if (a.nwords == 1) {
      if (n_cand > 1) pa.mask[tile] = m;
      any |= (m != 0);
    } else {
      while (cur < a.nwords) { pa.mask[tile * a.nwords + cur] = m; any |= (m != 0); m = 0; ++cur; }
    }
  }

  // compact the tile entries into this block's own list segments.  The "has candidates"
  // segment is ordered by cost class -- 3+ candidates, 2, 1 -- so that consumers can deal and
  // schedule by class (class 3 = no candidate -> the "none" segment)
  const int cls = !valid ? 4 : (n_cand >= 3 ? 0 : (n_cand == 2 ? 1 : (n_cand == 1 ? 2 : 3)));
  const unsigned long long lt = (1ull << lane) - 1ull;
  unsigned long long mine = 0;                   // ballot of this thread's class
  unsigned long long wave_n = 0;                 // 4 x 16 bit: this wave's count per class
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const unsigned long long bal = __ballot(cls == c);
    if (cls == c) mine = bal;
    wave_n |= (unsigned long long)__builtin_popcountll(bal) << (16 * c);
  }
  __shared__ unsigned long long s_cls_n[4];
  if (lane == 0) s_cls_n[wave] = wave_n;
  __syncthreads();
  unsigned long long pre = 0, tot = 0;           // 4 x 16 bit: earlier waves / whole block
  for (int w = 0; w < 4; ++w) {
    const unsigned long long x = s_cls_n[w];
    if (w < wave) pre += x;
    tot += x;
  }
  const unsigned n0 = (unsigned)tot & 0xffffu, n1 = (unsigned)(tot >> 16) & 0xffffu, n2 = (unsigned)(tot >> 32) & 0xffffu,
                 n3 = (unsigned)(tot >> 48);
  if (tid == 0) {
    pa.blk_counts[2 * blockIdx.x] = n0 | (n1 << 16);
    pa.blk_counts[2 * blockIdx.x + 1] = n2 | (n3 << 16);
  }
  if (valid) {
    const unsigned long long e = encode_entry(tx, ty, b, k_first, n_cand > 1);
    const size_t seg = (size_t)blockIdx.x * SEG;
    const unsigned rank = ((unsigned)(pre >> (16 * cls)) & 0xffffu) + __builtin_popcountll(mine & lt);
    const unsigned start = cls == 0 ? 0u : (cls == 1 ? n0 : (cls == 2 ? n0 + n1 : 0u));
    (cls == 3 ? pa.empty : pa.heavy)[seg + start + rank] = e;
  }
}

// k_fwd's image stores: buffer_store_dwordx3 through a per-tile buffer resource -- the tile's first row as the base, the
// row as a scalar offset, the lane's 12-byte slot as the vector offset: no address arithmetic on the VALU -- WRITTEN THROUGH
// (sc1).  A plain store leaves its line dirty in the XCD's L2 and the end of the kernel writes back what has piled up
// there; written through, the lines leave L2 as they come: a kernel of nothing but these stores takes 24.6 instead of 26.2 us
// (tools/ubench/store_patterns.hip; nt stores 37 us), k_fwd 31.0 instead of 32.0 (three A/B rounds on one box).
#ifndef RRT_FWD_STORE_AUX
#define RRT_FWD_STORE_AUX 16          // cache-policy bits of the store: 0 plain, 16 sc1, 17 sc0 sc1, 2 nt
#endif
typedef float v3f __attribute__((ext_vector_type(3)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(char* first_row) {
  // raw buffer (stride 0), no real bound (a tile's stores stay within P rows of its base), gfx9 32-bit float format word
  return __builtin_amdgcn_make_buffer_rsrc(first_row, 0, 0x7fffffff, 0x27000);
}
__device__ __forceinline__ void store_rgb(__amdgpu_buffer_rsrc_t rs, unsigned lane_off, unsigned row_off, float r, float g, float b) {
  const v3f v = {r, g, b};
  __builtin_amdgcn_raw_buffer_store_b96(v, rs, (int)lane_off, (int)row_off, RRT_FWD_STORE_AUX);
}


// First row of tile t in an image / gradient buffer [B][rows][ncols][3], as a wave-uniform byte pointer.
__device__ __forceinline__ char* tile_row_ptr(const float* buf, const Tile& t) {
  ckargs ka = kargs_here();
  const size_t row_bytes = (size_t)ka->v.ncols * 12;
  return reinterpret_cast<char*>(const_cast<float*>(buf)) + ((size_t)t.b * ka->v.rows + t.a0) * row_bytes + (size_t)t.b0 * 12;
}

// ---- per-workgroup work queue ------------------------------------------------
// Workgroup wg owns entries wg, wg + n_wgs, ... of a tile list (a fair, spatially
// interleaved share).  They are copied into LDS once and handed to the
// workgroup's 16 waves through an LDS cursor (ds_add_rtn: ~100 cycles, no global
// atomics): a tile costs 1x..3x depending on its candidates and coverage, and a
// static deal per WAVE left most SIMDs idle behind the unluckiest wave
// (profiles/: 2.6 of 4 resident waves busy on average).
#ifndef RRT_EARLY
#define RRT_EARLY 32
#endif
constexpr int EARLY = RRT_EARLY;       // "none" slots per segment that its owner workgroup zero-fills in the prologue
constexpr int Q_CAP = 1024;            // entries staged per batch and workgroup (= its threads)
struct WgQueue {
  unsigned long long ent[Q_CAP];
  unsigned scan[4 * WAVES];
  unsigned ex[3][Q_CAP];           // per cost class: exclusive prefix of the chunk's segment sizes
  unsigned cnt01[Q_CAP];           // segment sizes of classes 0 and 1 (16 bits each)
  unsigned cursor;
};
// k_fwd only: the workgroup's share of the tiles without candidates (zero-filled)
struct WgZeroQueue {
  unsigned long long ent[Q_CAP];
  unsigned ex[Q_CAP];
  unsigned long long early[(Q_CAP / SEG) * EARLY];   // the head of the workgroup's own segments (see zero_early_load)
  unsigned cursor;
};

// Inclusive prefix sum over the 64 lanes of a wave: Hillis-Steele inside each row of 16
// (row_shr 1, 2, 4, 8), then row totals forwarded with row_bcast:15 / row_bcast:31.  Lanes a
// DPP control does not reach add 0.  Six VALU adds, no LDS.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_add_u32(unsigned x) {
  return x + (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ unsigned wave_incl_scan(unsigned x) {
  x = dpp_add_u32<0x111, 0xf>(x);   // row_shr:1
  x = dpp_add_u32<0x112, 0xf>(x);   // row_shr:2
  x = dpp_add_u32<0x114, 0xf>(x);   // row_shr:4
  x = dpp_add_u32<0x118, 0xf>(x);   // row_shr:8
  x = dpp_add_u32<0x142, 0xa>(x);   // row_bcast:15 into rows 1,3
  x = dpp_add_u32<0x143, 0xc>(x);   // row_bcast:31 into rows 2,3
  return x;
}

// Exclusive prefix sums of N <= 4 values per thread over the workgroup's 16 waves, and the
// grand totals (wave-uniform).  One barrier; s_w[N*16] must not be in use.  The 16 wave
// totals of value n land in row n of ONE register (lane 16n + w), so the scan over waves is
// four DPP adds and the results are picked with v_readlane -- a loop over LDS words here cost
// 3 us of the kernels' prologue.
template <int NW, int N>
__device__ __forceinline__ void block_excl_scan(const unsigned (&x)[N], unsigned* s_w, int lane, int wave,
                                                unsigned (&ex)[N], unsigned (&tot)[N]) {
  static_assert(NW == 16 && N <= 4, "one 16-lane row per value");
  unsigned incl[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    incl[n] = wave_incl_scan(x[n]);
    if (lane == 63) s_w[16 * n + wave] = incl[n];
  }
  __syncthreads();
  const unsigned own = lane < 16 * N ? s_w[lane] : 0u;
  unsigned v = own;
  v = dpp_add_u32<0x111, 0xf>(v);   // row_shr:1
  v = dpp_add_u32<0x112, 0xf>(v);   // row_shr:2
  v = dpp_add_u32<0x114, 0xf>(v);   // row_shr:4
  v = dpp_add_u32<0x118, 0xf>(v);   // row_shr:8 -> inclusive over the waves, per row
  const unsigned before = v - own;
#pragma unroll
  for (int n = 0; n < N; ++n) {
    tot[n] = (unsigned)__builtin_amdgcn_readlane((int)v, 16 * n + 15);
    ex[n] = (unsigned)__builtin_amdgcn_readlane((int)before, 16 * n + wave) + incl[n] - x[n];
  }
}

// All 1024 threads: k_prep left every block's entries in the block's own segment, ordered by
// cost class (3+ candidates, 2, 1).  Thread t brings cw = the sizes of segment bin0 + t of the
// chunk [bin0, bin0 + nb).  The chunk's (virtual) list is "every segment's class 0, then every
// segment's class 1, then class 2": three workgroup scans give each entry its index in it, and
// the workgroup owns the indices blockIdx.x, + gridDim.x, ... -- an equal share of every
// class, staged expensive first (longest-processing-time-first inside the workgroup).
// QueueState carries the first index not yet staged and the number of entries before the
// chunk.  Returns the number staged (<= Q_CAP: the caller sizes the chunk for that).  k_fwd
// (WITH_EMPTY) stages its share of the "none" list the same way into the zero queue.
// Which indices of the "none" list a workgroup of k_fwd zero-fills.  The store path of the odd XCDs of an MI355X is
// ~15 % slower than that of the even ones (tools/ubench/xcc_speed.hip: 21-23 us against 18.4 us for the same 768 KB per
// workgroup; loads and VALU within 2 %), blocks go to XCDs round-robin (block b on XCD b mod 8 on every box seen), and
// k_fwd's store stream is bound per XCD: with equal shares the odd XCDs' workgroups finished 2.2 us after the even ones
// (tools/stamp_fwd.py).  So the list is dealt in rounds of w_odd full rows -- one index per workgroup -- followed by
// w_even - w_odd half rows for the workgroups with even blockIdx only: they zero-fill w_even tiles for every w_odd of an odd
// one.  The partition depends on blockIdx alone: were blocks placed differently, only the balance would suffer.
// w_even = w_odd is the plain interleaved deal (index j n + me).
struct XcdDeal {
  unsigned n, me, w_even, w_odd, round;
  __device__ __forceinline__ XcdDeal(unsigned n_, unsigned me_, unsigned we, unsigned wo) : n(n_), me(me_), w_even(we), w_odd(wo) {
    if ((n & 1u) || w_even < w_odd || w_odd == 0u) w_even = w_odd = 1u;
    round = w_odd * n + (w_even - w_odd) * (n / 2);
  }
  __device__ __forceinline__ unsigned mine() const { return (me & 1u) ? w_odd : w_even; }   // indices per round
  // the j-th index this workgroup owns
  __device__ __forceinline__ unsigned owned(unsigned j) const {
    const unsigned w = mine(), r = j / w, k = j - r * w;
    return r * round + (k < w_odd ? k * n + me : w_odd * n + (k - w_odd) * (n / 2) + me / 2);
  }
  // how many of its indices lie below x
  __device__ __forceinline__ unsigned below(unsigned x) const {
    const unsigned r = x / round, rem = x - r * round;
    unsigned c = r * mine();
    if (rem > me) c += min(w_odd, (rem - me + n - 1) / n);
    if (!(me & 1u) && rem > w_odd * n + me / 2) c += min(w_even - w_odd, (rem - w_odd * n - me / 2 + n / 2 - 1) / (n / 2));
    return c;
  }
};

struct QueueState {
  unsigned base = 0, next = 0;       // "has candidates": entries before the chunk / first index not yet staged
  unsigned base_e = 0, j_e = 0;      // "none" (k_fwd only): entries before the chunk / how many of its indices are staged
  unsigned cnt_e = 0;                // "none" entries staged in the current batch
};

struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `mid` runs between the scans and the entry gather (k_fwd issues its first zero-fill stores there).
template <bool WITH_EMPTY, class Hook>
__device__ __forceinline__ unsigned queue_fill(WgQueue& q, WgZeroQueue* z, const unsigned long long* heavy,
                                               const unsigned long long* empty, uint2 cw, unsigned bin0, unsigned nb,
                                               QueueState& st, int lane, int wave, Hook mid,
                                               unsigned long long* save, unsigned* save_cnt, unsigned zw_even = 1u,
                                               unsigned zw_odd = 1u) {
  constexpr int N = WITH_EMPTY ? 4 : 3;
  const unsigned t = threadIdx.x, n = gridDim.x;
  unsigned c[N], ex[N], tot[N];
  c[0] = cw.x & 0xffffu; c[1] = cw.x >> 16; c[2] = cw.y & 0xffffu;
  if (WITH_EMPTY) c[N - 1] = (cw.y >> 16) > EARLY ? (cw.y >> 16) - EARLY : 0u;   // the owner takes the first EARLY
  block_excl_scan<WAVES, N>(c, q.scan, lane, wave, ex, tot);
  q.ex[0][t] = ex[0]; q.ex[1][t] = ex[1]; q.ex[2][t] = ex[2];
  q.cnt01[t] = cw.x;
  if (WITH_EMPTY) z->ex[t] = ex[N - 1];
  if (t == 0) { q.cursor = 0; if (WITH_EMPTY) z->cursor = 0; }
  __syncthreads();
  { const int gw = blockIdx.x * WAVES + wave; (void)gw; RRT_STAMP_FWD(6, ~0ull); }
  mid();
  const unsigned end = st.base + tot[0] + tot[1] + tot[2];
  const unsigned end_e = WITH_EMPTY ? st.base_e + tot[N - 1] : 0u;
  const XcdDeal zd(n, blockIdx.x, zw_even, zw_odd);
  const unsigned g = st.next + t * n, ge = WITH_EMPTY ? zd.owned(st.j_e + t) : 0u;
  const bool on = g < end, on_e = WITH_EMPTY && ge < end_e;
  if (__ballot(on || on_e)) {
    unsigned x = on ? g - st.base : 0u;
    const int cls = x < tot[0] ? 0 : (x < tot[0] + tot[1] ? 1 : 2);
    x -= cls == 0 ? 0u : (cls == 1 ? tot[0] : tot[0] + tot[1]);
    const unsigned xe = on_e ? ge - st.base_e : 0u;
    const unsigned* pre = q.ex[cls];
    unsigned lo = 0, le = 0;       // both binary searches side by side: their LDS reads overlap
#pragma unroll
    for (unsigned step = Q_CAP / 2; step; step >>= 1) {
      if (step >= nb) continue;        // (wave-uniform: a step of nb or more cannot stay below nb)
      const unsigned i = lo + step, ie = le + step;
      const unsigned pv = pre[min(i, nb - 1)];
      if (WITH_EMPTY) {
        const unsigned ev = z->ex[min(ie, nb - 1)];
        if (ev <= xe && ie < nb) le = ie;
      }
      if (pv <= x && i < nb) lo = i;
    }
    { const int gw = blockIdx.x * WAVES + wave; (void)gw; RRT_STAMP_FWD(7, ~0ull); }
    if (on) {
      const unsigned c01 = q.cnt01[lo];
      const unsigned start = cls == 0 ? 0u : (cls == 1 ? (c01 & 0xffffu) : (c01 & 0xffffu) + (c01 >> 16));
      const unsigned long long e = heavy[(size_t)(bin0 + lo) * SEG + start + (x - pre[lo])];
      q.ent[t] = e;
      save[t] = e;
    }
    if (on_e) z->ent[t] = empty[(size_t)(bin0 + le) * SEG + EARLY + (xe - z->ex[le])];
  }
  const unsigned cnt = st.next < end ? (end - st.next + n - 1) / n : 0u;
  if (t == 0) *save_cnt = cnt;
  st.next += cnt * n;
  st.base = end;
  if (WITH_EMPTY) {
    st.cnt_e = zd.below(end_e) - st.j_e;
    st.j_e += st.cnt_e;
    st.base_e = end_e;
  }
  __syncthreads();
  return cnt;
}

// k_fwd: HBM should not idle while the workgroups sort out their lists, so every segment
// sets its first EARLY "none" entries aside for its OWNER (segment bin0 + j belongs to
// workgroup j mod gridDim.x): no scan, no search, a contiguous read that is in flight from
// the first instruction of the kernel, and the zero-fill stores leave ~1.5 us into it.  Only
// the rest of a segment's "none" entries is dealt through the balanced queue (a workgroup
// left to zero-fill ALL its own segments ran 5 us behind the others).
__device__ __forceinline__ unsigned long long zero_early_load(const KArgs& a, unsigned bin0, unsigned nb) {
  const unsigned n = gridDim.x, t = threadIdx.x;
  if (t >= (Q_CAP / SEG) * EARLY) return ~0ull;
  const unsigned j = (blockIdx.x + n - bin0 % n) % n + (t / EARLY) * n;   // chunk-relative own segment
  if (j >= nb) return ~0ull;
  const unsigned size = a.blk_counts[2 * (bin0 + j) + 1] >> 16;
  const unsigned long long e = a.empty[(size_t)(bin0 + j) * SEG + t % EARLY];
  return t % EARLY < size ? e : ~0ull;
}

__device__ __forceinline__ bool entry_is_multi(unsigned long long e) { return ((e >> 62) & 1u) != 0; }

// One wave: next entry of the batch, or false when it is exhausted.
__device__ __forceinline__ bool queue_pop(WgQueue& q, unsigned cnt, int lane, unsigned long long& e) {
  unsigned i = 0;
  if (lane == 0) i = atomicAdd(&q.cursor, 1u);
  i = __builtin_amdgcn_readfirstlane(i);
  if (i >= cnt) return false;
  e = q.ent[i];
  e = lane_bcast64(e, 0);
  return true;
}

// k_bwd: the same, every entry handed out 1 << split_log2 times (consecutively: the waves that share a tile run together and
// find its gradient rows in the L2); `part` says which share of the samples the caller got.  SP = 0: queue_pop.
template <int SP>
__device__ __forceinline__ bool queue_pop_part(WgQueue& q, unsigned cnt, int lane, unsigned long long& e, int& part) {
  if (!SP) { part = 0; return queue_pop(q, cnt, lane, e); }
  const int sl = kargs_here()->split_log2;
  unsigned i = 0;
  if (lane == 0) i = atomicAdd(&q.cursor, 1u);
  i = __builtin_amdgcn_readfirstlane(i);
  if (i >= (cnt << sl)) return false;
  e = q.ent[i >> sl];
  e = lane_bcast64(e, 0);
  part = (int)(i & ((1u << sl) - 1u));
  return true;
}

#ifndef RRT_BWD_WAVES_PER_SIMD
#define RRT_BWD_WAVES_PER_SIMD 1
#endif
#ifndef RRT_FWD_WAVES_PER_SIMD
#define RRT_FWD_WAVES_PER_SIMD 4
#endif

// Colour of one winning hit: FAST_* shapes (rrt_device.h) take the lean Phong path (wave-uniform branch).
template <int SHADER, int SPEC>
__device__ __forceinline__ void shade_hit(const KArgs& a, const ShapeP& q, const SceneP& sc, float tmin,
                                          const float nu[3], float x[3]) {
  if (SHADER == SHADER_DEPTH) {
    x[0] = x[1] = x[2] = shade_depth<float>(tmin, a.v.inv_max_depth);
  } else if (q.powf & (SPEC ? FAST_SPEC : FAST_NOSPEC)) {
    shade_phong_fast<float, SPEC>(nu, q.colI, q.ks, q.kd, q.ka, q.sh, q.powf, sc.L, sc.look, sc.Llook, x);
  } else {
    shade_phong<float>(nu, q.colI, q.ks, q.kd, q.ka, q.sh, q.powf, SPEC, sc.L, sc.look, x);
  }
}

// ---------------------------------------------------------------- forward --
// Persistent, one 1024-thread workgroup per CU.  The workgroup's share of the "has
// candidates" list (an equal part of every cost class, expensive first) and of the "none"
// list sit in two LDS queues; a wave pops a tile to render and, with it, its part of the
// tiles to zero-fill, so the store stream runs under the arithmetic and no wave idles
// behind an unlucky static deal (a wave timeline -- tools/stamp_fwd.py -- showed a third of
// a statically dealt wave's life in the list prologue and an 11 us tail).
template <int P, int SHADER, int SPEC, int JC, int TR, int SH>
__global__ __launch_bounds__(BLOCK, RRT_FWD_WAVES_PER_SIMD) void k_fwd(KArgs a) {
  __shared__ float s_scene[LDS_RECS * PS_STRIDE];
  __shared__ WgQueue s_q;
  __shared__ WgZeroQueue s_z;
  const ViewK& v = a.v;
  int lane;
  const int gw = global_wave<WAVES>(lane);
  const int wave = gw - (int)blockIdx.x * WAVES;
  // chunk of list segments per batch: one per thread, and no more than Q_CAP tiles for this workgroup
  const unsigned chunk = (unsigned)a.chunk;
  RRT_STAMP_FWD(0, ~0ull);
  // (the segment sizes are in flight under the staging)
  uint2 cw = threadIdx.x < min(chunk, (unsigned)a.n_bins) ? reinterpret_cast<const uint2*>(a.blk_counts)[threadIdx.x]
                                                           : make_uint2(0u, 0u);
  unsigned long long early = zero_early_load(a, 0u, min(chunk, (unsigned)a.n_bins));
  stage_scene(a, s_scene);          // (ends with the workgroup barrier)
  RRT_STAMP_FWD(5, ~0ull);

  const unsigned lane12 = (unsigned)lane * 12u;
  auto zero_tile = [&](unsigned long long e) {
    int k0; bool multi; long long tile;
    const Tile t = decode_entry(a, e, k0, multi, tile);
    // (the tile's first row as a buffer resource, the row a scalar offset, the lane's 12-byte slot the vector offset; as
    // `image + (...) * 3` with the lane folded into a 64-bit vector base, every row cost two v_mad_u64_u32)
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(tile_row_ptr(a.image, t));
    const unsigned row_bytes = (unsigned)kargs_here()->v.ncols * 12u;
    if (lane < t.ncols) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (p < t.nrows) store_rgb(rs, lane12, p * row_bytes, 0.f, 0.f, 0.f);
      }
    }
  };
  // take `count` entries off the zero queue; false once it is exhausted
  auto zero_fill = [&](unsigned count, unsigned cnt_e) {
    unsigned i = 0;
    if (lane == 0) i = atomicAdd(&s_z.cursor, count);
    i = __builtin_amdgcn_readfirstlane(i);
    const unsigned hi = min(i + count, cnt_e);
    for (unsigned n = i; n < hi; ++n) zero_tile(lane_bcast64(s_z.ent[n], 0));
    return i + count < cnt_e;
  };

  // the head of the workgroup's own "none" segments: stores that need no list work, issued between the scans and the
  // searches of the deal (right behind the staging barrier, 1.5 us earlier, they made the kernel 0.6 us SLOWER: measured)
  auto zero_early = [&] {
    for (unsigned i = wave; i < (Q_CAP / SEG) * EARLY; i += WAVES) {
      const unsigned long long e = lane_bcast64(s_z.early[i], 0);
      if (e != ~0ull) zero_tile(e);
    }
  };
  int b_cur = -1;
  SceneP sc = {{0.f, 0.f, 1.f}, {0.f, 0.f, 1.f}, 1.f};
  QueueState qs;
  qs.next = blockIdx.x;
  // lds_scene = 2 (a batch of many small scenes): no global deal -- the workgroup takes whole list segments
  // (its own, blockIdx.x + i * gridDim.x), stages the segment's scene and works through the segment's entries
  const bool local = a.lds_scene == 2;
  for (unsigned bin0 = local ? blockIdx.x : 0u; bin0 < (unsigned)a.n_bins; bin0 += local ? gridDim.x : chunk) {
    unsigned cnt, cnt_e;
    if (local) {
      if (bin0 != blockIdx.x) __syncthreads();               // previous segment fully consumed
      const uint2 c = reinterpret_cast<const uint2*>(a.blk_counts)[bin0];
      cnt = (c.x & 0xffffu) + (c.x >> 16) + (c.y & 0xffffu);
      cnt_e = c.y >> 16;
      stage_one_scene(a, s_scene, (int)(bin0 / (unsigned)a.bins_per_scene));
      if (threadIdx.x < cnt) s_q.ent[threadIdx.x] = a.heavy[(size_t)bin0 * SEG + threadIdx.x];
      if (threadIdx.x < cnt_e) s_z.ent[threadIdx.x] = a.empty[(size_t)bin0 * SEG + threadIdx.x];
      if (threadIdx.x == 0) { s_q.cursor = 0; s_z.cursor = 0; }
      __syncthreads();
    } else {
      const unsigned nb = min(chunk, (unsigned)a.n_bins - bin0);
      if (bin0) {
        __syncthreads();                     // previous batch fully consumed
        cw = threadIdx.x < nb ? reinterpret_cast<const uint2*>(a.blk_counts)[bin0 + threadIdx.x] : make_uint2(0u, 0u);
        early = zero_early_load(a, bin0, nb);
      }
      if (threadIdx.x < (Q_CAP / SEG) * EARLY) s_z.early[threadIdx.x] = early;
      const size_t slot = (size_t)(bin0 / chunk) * gridDim.x + blockIdx.x;
      cnt = queue_fill<true>(s_q, &s_z, a.heavy, a.empty, cw, bin0, nb, qs, lane, wave, zero_early,
                             a.wgq + slot * Q_CAP, a.wgq_cnt + slot, (unsigned)a.zw_even, (unsigned)a.zw_odd);
      cnt_e = qs.cnt_e;
    }
    // zero-filled tiles handed out with every rendered tile.  Not all of them: a share (a.ztail per cent) stays behind for
    // the waves that run out of tiles to render -- a zero-filled tile is 0.1 us of work, the ideal filler for the 2-3 us
    // between the first and the last wave of a workgroup leaving its render loop
    const unsigned per = cnt ? (cnt_e * (100u - (unsigned)a.ztail) / 100u + cnt - 1) / cnt : 0;
    RRT_STAMP_FWD(1, ~0ull);
    unsigned long long e_cur;
    bool more_zero = cnt_e > 0;
    // one tile; MULTI = several candidate shapes.  The queue is ordered by cost class, so a wave renders its
    // multi-candidate tiles in one loop and the single-candidate ones in a second, leaner one
    auto render_tile = [&](auto multi_tag) {
      constexpr bool MULTI = decltype(multi_tag)::value;
      int k0; bool multi_bit; long long tile;
      const Tile t = decode_entry(a, e_cur, k0, multi_bit, tile);
      (void)multi_bit;
      if (t.b != b_cur) {
        load_scene(a, t.b, sc);
        b_cur = t.b;
      }
      cuptr mk = as_const(kargs_here()->mask) + tile * kargs_here()->nwords;
      const LaneRay lr = lane_ray<TR>(a, t, lane);
      float c[P][3];
#pragma unroll
      for (int p = 0; p < P; ++p) c[p][0] = c[p][1] = c[p][2] = 0.f;
#pragma unroll 1
      for (int s = 0; s < v.A; ++s) {
        float d[P][3], tmin[P], nu[P][3];
        int best[P];
        gen_rays<P, JC, TR>(a, t, lr, s, d);
#pragma unroll
        for (int p = 0; p < P; ++p) { best[p] = -1; tmin[p] = INFINITY; }
        if (!MULTI) {
          // one candidate: its record is read once and serves trace and shading
          // (its transform as per-lane VGPR copies: VALU instructions with an SGPR source issue at half rate)
          ShapeP q;
          load_shape(a, s_scene, t.b, k0, lane, q);
          float Av[9], ov[3];
          load_shape_vgpr(a, s_scene, t.b, k0, Av, ov);
          trace_shape_p<P, SHADER == SHADER_PHONG>(q.kind, Av, ov, k0, d, best, tmin, nu);
          if (SH) {
#pragma unroll
            for (int p = 0; p < P; ++p)
              if (wave_in_shadow(a, s_scene, t.b, best[p], d[p], tmin[p], best[p] >= 0, lane)) best[p] = -1;   // black
          }
#pragma unroll
          for (int p = 0; p < P; ++p) {
            if (best[p] >= 0) {
              float x[3];
              shade_hit<SHADER, SPEC>(a, q, sc, tmin[p], nu[p], x);
              c[p][0] += x[0]; c[p][1] += x[1]; c[p][2] += x[2];
            }
          }
        } else {
          for (int w = 0; w < a.nwords; ++w) {
            unsigned long long m = mk[w];
            while (m) {
              int k = w * 64 + __builtin_ctzll(m);
              m &= m - 1;
              ShapeP q;
              load_shape(a, s_scene, t.b, k, lane, q);
              trace_shape<P, SHADER == SHADER_PHONG>(q, k, d, best, tmin, nu);
            }
          }
          if (SH) {
#pragma unroll
            for (int p = 0; p < P; ++p)
              if (wave_in_shadow(a, s_scene, t.b, best[p], d[p], tmin[p], best[p] >= 0, lane)) best[p] = -1;   // black
          }
          if (SHADER == SHADER_DEPTH) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
              if (best[p] >= 0) {
                float x = shade_depth<float>(tmin[p], v.inv_max_depth);
                c[p][0] += x; c[p][1] += x; c[p][2] += x;
              }
            }
          } else {
            // shade shape by shape: material in SGPRs, only the lanes it won are active
            for (int w = 0; w < a.nwords; ++w) {
              unsigned long long m = mk[w];
              while (m) {
                int k = w * 64 + __builtin_ctzll(m);
                m &= m - 1;
                bool mine = false;
#pragma unroll
                for (int p = 0; p < P; ++p) mine |= (best[p] == k);
                if (!__ballot(mine)) continue;
                ShapeP q;
                load_shape(a, s_scene, t.b, k, lane, q);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                  if (best[p] == k) {
                    float x[3];
                    shade_hit<SHADER_PHONG, SPEC>(a, q, sc, tmin[p], nu[p], x);
                    c[p][0] += x[0]; c[p][1] += x[1]; c[p][2] += x[2];
                  }
                }
              }
            }
          }
        }
      }
      // one dwordx3 per lane and row: the wave writes 768 contiguous bytes.  (The empty asm
      // keeps the address arithmetic down here: hoisted above the sample loop it was spilled.)
      int a0 = t.a0, b0 = t.b0;
      asm volatile("" : "+s"(a0), "+s"(b0));
      Tile ts = t;
      ts.a0 = a0; ts.b0 = b0;
      const __amdgpu_buffer_rsrc_t rs = tile_rsrc(tile_row_ptr(a.image, ts));      // (as in zero_tile)
      const unsigned row_bytes = (unsigned)kargs_here()->v.ncols * 12u;
      if (lane < t.ncols) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if (p < t.nrows) store_rgb(rs, lane12, p * row_bytes, c[p][0] * v.inv_A, c[p][1] * v.inv_A, c[p][2] * v.inv_A);
        }
      }
    };
    bool have = queue_pop(s_q, cnt, lane, e_cur);
#pragma unroll 1
    while (have && entry_is_multi(e_cur)) {
      if (more_zero) more_zero = zero_fill(per, cnt_e);
      render_tile(std::true_type());
      have = queue_pop(s_q, cnt, lane, e_cur);
    }
#pragma unroll 1
    while (have) {
      if (more_zero) more_zero = zero_fill(per, cnt_e);
      render_tile(std::false_type());
      have = queue_pop(s_q, cnt, lane, e_cur);
    }
    while (more_zero) more_zero = zero_fill(2, cnt_e);
  }
  RRT_STAMP_FWD(2, ~0ull);
  // (diagnostic builds) the XCD this wave ran on: HW_REG_XCC_ID[3:0]
  RRT_STAMP_FWD(3, 1 + (__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15));
}

// --------------------------------------------------------------- backward --
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float x) {
  int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false);
  return x + __int_as_float(moved);
}

// Sum over the 64 lanes of a wave; the total lands in lane 63 (all lanes must
// be active).  quad_perm, quad_perm, row_shr:4, row_shr:8, row_bcast:15,
// row_bcast:31 -- six full-rate VALU adds, no LDS.
__device__ __forceinline__ float wave_sum_to_lane63(float x) {
  x = dpp_add<0xb1, 0xf>(x);    // quad_perm:[1,0,3,2]
  x = dpp_add<0x4e, 0xf>(x);    // quad_perm:[2,3,0,1]
  x = dpp_add<0x114, 0xf>(x);   // row_shr:4
  x = dpp_add<0x118, 0xf>(x);   // row_shr:8
  x = dpp_add<0x142, 0xa>(x);   // row_bcast:15 into rows 1,3
  x = dpp_add<0x143, 0xc>(x);   // row_bcast:31 into rows 2,3
  return x;
}

// acc[0..21] (per-lane partial sums, all 64 lanes active) -> r6: row j (lanes 16j..16j+15) of
// r6[m], m < 5, ends with total m + {0, 6, 11, 17}[j] in its last lane; r6[5] with total 5 in
// lane 15 and total 16 in lane 47.
__device__ __forceinline__ void wave_sums_22(const float acc[ACC_USED], float r6[6]) {
  static_assert(ACC_USED == 22, "the butterfly is laid out for 22 values");
  float r[11];
#pragma unroll
  for (int n = 0; n < 11; ++n) {
    // rows 2,3 of acc[n] <-> rows 0,1 of acc[n+11]: the sum folds lane i onto lane i^32,
    // value n in the lower half of the wave, value n+11 in the upper
    auto pr = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[n]), __float_as_uint(acc[n + 11]), false, false);
    r[n] = __uint_as_float(pr[0]) + __uint_as_float(pr[1]);
  }
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    // odd rows of r[m] <-> even rows of r[m+6]: rows 0,2 keep r[m]'s two values, rows 1,3 get r[m+6]'s
    auto pr = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[m]), __float_as_uint(r[m < 5 ? m + 6 : m]), false, false);
    float x = __uint_as_float(pr[0]) + __uint_as_float(pr[1]);
    x = dpp_add<0xb1, 0xf>(x);    // quad_perm:[1,0,3,2]
    x = dpp_add<0x4e, 0xf>(x);    // quad_perm:[2,3,0,1]
    x = dpp_add<0x114, 0xf>(x);   // row_shr:4
    x = dpp_add<0x118, 0xf>(x);   // row_shr:8 -> the row's total in its last lane
    r6[m] = x;
  }
}

// The upstream-gradient rows of a tile go straight from HBM into the wave's LDS staging buffer, one
// tile ahead of their use (LDS-DMA: no VGPRs, no ds_write).  A tile row is ncols * 12 contiguous
// bytes: lanes 0..n16-1 move 16 bytes each (global_load_lds_dwordx4 lands lane l's bytes at LDS
// offset 16 l, so the row arrives compact, pixel-major [pixel][3]) and, on ragged tiles only, up to
// three lanes move the remaining dwords one by one -- no byte beyond the row is touched.  Pixel l
// reads words 3l..3l+2: conflict-free, 3 is odd.  Rows outside the image are not loaded (their slots
// keep stale bytes; the row loops never read them).  The compiler does not track the asynchronous
// LDS write: the consumer waits with s_waitcnt vmcnt(0).
// (global_load_lds_dwordx3 pads every lane to 16 bytes in LDS -- measured -- which would make the
// double buffer 128 KB.)
constexpr int UP_ROW = 64 * 3;
template <int P>
__device__ __forceinline__ void prefetch_grad_rows(const KArgs& a, const Tile& t, int lane_unused, float* dst) {
  typedef const __attribute__((address_space(1))) void* gptr;
  typedef __attribute__((address_space(3))) void* lptr;
  // the lane id straight from the hardware (two VALU ops): derived from threadIdx it stays live across the
  // whole tile loop as a 64-bit address offset, and the register allocator spilled it in front of every load
  (void)lane_unused;
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  // one address computation per tile, one add per row
  ckargs ka = kargs_here();
  const size_t row_bytes = (size_t)ka->v.ncols * 12;
  const char* row = reinterpret_cast<const char*>(ka->grad) + ((size_t)t.b * ka->v.rows + t.a0) * row_bytes + (size_t)t.b0 * 12;
  if (t.ncols == TILE_W) {
    // the usual case, a full-width tile: 48 lanes x 16 bytes per row
    if (lane < TILE_W * 3 / 4) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (p < t.nrows)
          __builtin_amdgcn_global_load_lds((gptr)(row + p * row_bytes + 16 * lane), (lptr)(dst + p * UP_ROW), 16, 0, 0);
      }
    }
  } else {
    const int nd = t.ncols * 3, n16 = nd >> 2, tail = nd & 3;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (p < t.nrows) {
        const char* r = row + p * row_bytes;
        if (lane < n16) __builtin_amdgcn_global_load_lds((gptr)(r + 16 * lane), (lptr)(dst + p * UP_ROW), 16, 0, 0);
        if (lane < tail)
          __builtin_amdgcn_global_load_lds((gptr)(r + 16 * n16 + 4 * lane), (lptr)(dst + p * UP_ROW + 4 * n16), 4, 0, 0);
      }
    }
  }
}

// A wave's position in its workgroup's queue, and what it carries from tile to tile.
struct BwdWave {
  unsigned long long e_cur, e_nxt;   // current entry, next entry (its gradient rows are in flight)
  bool has_cur;
  int part_cur, part_nxt;            // (SP) which share of the tile's samples this wave works on
  int buf;                           // which half of the wave's gradient double buffer holds the current tile's rows
  int b_cur;                         // scene whose constants `sc` holds
  SceneP sc;
  // (only when the scene does not fit the LDS accumulators) the last global accumulator flush of a
  // tile is held back into the next iteration: issued at the end of its tile it would sit in front of
  // the next tile's wait for its (older) gradient rows -- the vector-memory counter retires in order
  bool held;
  int held_k, held_b;
  float held_val;
};

// The GENERIC adjoint of one shape over one tile -- squares, the depth shader, dark or negative colours, exponents
// <= 1 -- as a function of its own (centre-jitter kernels only, see its call).  Inlined as a branch beside the hot path
// it shares the tile body's register allocation: the hot path (FAST_* spheres, all of the bench scene) ran with 41
// spilled vector registers and scratch accesses; without that branch it needs 122 registers and none (k_bwd 38.0 ->
// 36.1 us), with the branch as a call 123 and none, but calls pin scalar values to the callee-saved registers: 36.8 us.
// The 22 sums come back by value (a stack slot), the kernel arguments are re-read from the kernarg segment; both are
// noise beside the ~260 VALU instructions of a generic row.
struct GenericSums {
  float v[ACC_USED];
  int anyhit;
};
template <int P, int SHADER, int SPEC, int JC, int TR, bool MULTI>
__device__ __attribute__((noinline)) GenericSums bwd_generic_shape(int tb, int ta0, int tnrows, int k, int s_lo, int s_hi,
                                                                  float lr_cl, float lr_q, int lr_col, int col_ok_i,
                                                                  float qword, unsigned up_at, unsigned best_at,
                                                                  unsigned long long kernarg) {
  // (arguments arrive in vector registers: the wave-uniform ones are made scalar again.  The builtin returns int:
  // mind the sign extension when two halves are glued)
  auto uni = [](unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane(x); };
  tb = (int)uni(tb); ta0 = (int)uni(ta0); tnrows = (int)uni(tnrows); k = (int)uni(k); s_lo = (int)uni(s_lo); s_hi = (int)uni(s_hi);
  // the wave's gradient rows and winner table, as LDS offsets (generic pointers to LDS passed into a real function send
  // this compiler into "Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base")
  typedef __attribute__((address_space(3))) const float* lcf;
  typedef __attribute__((address_space(3))) const int* lci;
  const lcf up_rows = (lcf)(size_t)uni(up_at);
  const lci best = (lci)(size_t)uni(best_at);
  // the kernel's kernarg pointer comes as an argument: __builtin_amdgcn_kernarg_segment_ptr() is NULL in a real
  // function (tools/ubench/noinline_probe.hip)
  const ckargs ka = (ckargs)(((unsigned long long)uni((unsigned)(kernarg >> 32)) << 32) | uni((unsigned)kernarg));
  KArgs a;
  {
    static_assert(sizeof(KArgs) % 4 == 0, "copied as dwords");
    cu32ptr src = (cu32ptr)ka;
    unsigned words[sizeof(KArgs) / 4];
#pragma unroll
    for (unsigned i = 0; i < sizeof(KArgs) / 4; ++i) words[i] = src[i];
    __builtin_memcpy(&a, words, sizeof(KArgs));
  }
  const ViewK& v = a.v;
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const bool col_ok = col_ok_i != 0;
  LaneRay lr;
  lr.cl = lr_cl; lr.q = lr_q; lr.col = lr_col;
  SceneP sc;
  load_scene(a, tb, sc, ka);
  ShapeP q;
  shape_head(qword, q);
  shape_material(qword, q);
  shape_transform(qword, q);
  GenericSums out;
  float* acc = out.v;
#pragma unroll
  for (int n = 0; n < ACC_USED; ++n) acc[n] = 0.f;
  bool anyhit = false;
  __builtin_amdgcn_s_setprio(0);                     // (as in the hot path)
#pragma unroll 1
  for (int s = s_lo; s < s_hi; ++s) {
    float jcx, jcy;
    centre_jitter(v, s, jcx, jcy);
#pragma unroll 1
    for (int p = 0; p < tnrows; ++p) {
      const bool on = MULTI ? (best[p * 64 + lane] == k) : col_ok;
      if (on) {
        float d[3];
        row_ray<JC, TR>(a, lr, ta0 + p, s, jcx, jcy, d);
        const lcf ur = up_rows + p * UP_ROW + 3 * lane;
        const float up[3] = {ur[0], ur[1], ur[2]};
        anyhit |= adjoint_ray<float>(q.kind, q.A, q.o, q.colI, q.ks, q.kd, q.ka, q.sh, q.powf, SHADER, SPEC,
                                     v.inv_max_depth, sc.L, sc.look, d, up, acc);
      }
    }
  }
  __builtin_amdgcn_s_setprio(2);
  out.anyhit = anyhit ? 1 : 0;
  return out;
}

// One tile of the backward pass.  MULTI = the tile has several candidate shapes: per sample, every
// ray's winner is settled first and the adjoint then runs per winning shape.  !MULTI (two thirds of
// the bench scene's tiles, and the cheap ones to get right): one candidate, no winner search, the
// samples run INSIDE the accumulation, so the tile pays one butterfly.  A workgroup's queue is
// ordered by cost class (3+ candidates, 2, 1), so a wave runs its MULTI tiles in one loop and its
// single-candidate tiles in a second, leaner one (fewer live scalars: the general loop spills SGPRs).
template <int P, int SHADER, int SPEC, int JC, int TR, bool MULTI, int SH, int SP>
__device__ __forceinline__ void bwd_tile(const KArgs& a, BwdWave& w, WgQueue& s_q, unsigned cnt, const float* s_scene,
                                         float* s_acc, float* s_up_wave, int (*s_best_wave)[64], int lane) {
  const ViewK& v = a.v;
  int k0; bool multi_bit; long long tile;
  const Tile t = decode_entry(a, w.e_cur, k0, multi_bit, tile);
  (void)multi_bit;
  RRT_STAMP(0);
  RRT_STAMP_RT(14);
  if (t.b != w.b_cur) {
    load_scene(a, t.b, w.sc);
    w.b_cur = t.b;
  }
  const SceneP& sc = w.sc;
  const bool col_ok = lane < t.ncols;
  const LaneRay lr = lane_ray<TR>(a, t, lane);
  cuptr mk = as_const(kargs_here()->mask) + tile * kargs_here()->nwords;
  // (SP) this wave's share of the samples
  const int smp_lo = SP ? (w.part_cur * v.A) >> kargs_here()->split_log2 : 0;
  const int smp_hi = SP ? ((w.part_cur + 1) * v.A) >> kargs_here()->split_log2 : v.A;
  // several candidates: every ray's winner for one sample
  auto settle = [&](int s) {
    float d[P][3], tmin[P];
    int best[P];
    gen_rays<P, JC, TR>(a, t, lr, s, d);
#pragma unroll
    for (int p = 0; p < P; ++p) { best[p] = -1; tmin[p] = INFINITY; }
    for (int wd = 0; wd < a.nwords; ++wd) {
      unsigned long long m = mk[wd];
      while (m) {
        int k = wd * 64 + __builtin_ctzll(m);
        m &= m - 1;
        ShapeP q;
        shape_head(shape_word(a, s_scene, t.b, k, lane), q);
        float Av[9], ov[3];
        load_shape_vgpr(a, s_scene, t.b, k, Av, ov);
        trace_shape_p<P, false>(q.kind, Av, ov, k, d, best, tmin, nullptr);
      }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) s_best_wave[p][lane] = (col_ok && p < t.nrows) ? best[p] : -1;
  };
  // the first sample's winners need no gradient: settled BEFORE the wait for the rows.  (Every wave's first tile waits for
  // its rows together with the 4095 other waves of the chip -- a 12.6-MB burst, >= 2.5 us -- and the queue starts with the
  // multi-candidate tiles: k_bwd 31.1 -> 30.1 us)
  if (MULTI) settle(smp_lo);
  // (1) this tile's gradient rows, requested one tile ago, have landed in the wave's buffer once its
  // vector-memory counter is at zero (nothing younger is outstanding at this point)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  RRT_STAMP(1);
  const float* up_rows = s_up_wave + w.buf * (P * UP_ROW);
  // (2) the accumulator atomic held back from the previous tile
  if (w.held) {
    if (lane < ACC_USED)
      __hip_atomic_fetch_add(acc_copy(a) + ((size_t)w.held_b * a.S + w.held_k) * ACC_STRIDE + lane, w.held_val,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w.held = false;
  }
  // (3) pull the next tile and request its gradient rows
  const bool has_nxt = queue_pop_part<SP>(s_q, cnt, lane, w.e_nxt, w.part_nxt);
  if (has_nxt) {
    int k1; bool m1; long long tl1;
    const Tile tn = decode_entry(a, w.e_nxt, k1, m1, tl1);
    prefetch_grad_rows<P>(a, tn, lane, s_up_wave + (w.buf ^ 1) * (P * UP_ROW));
  }
  RRT_STAMP(2);

#pragma unroll 1
  for (int s0 = MULTI ? smp_lo : 0; s0 < (MULTI ? smp_hi : 1); ++s0) {
    const int s_lo = MULTI ? s0 : smp_lo, s_hi = MULTI ? s0 + 1 : smp_hi;
    if (MULTI && s0 != smp_lo) settle(s0);
    RRT_STAMP(3);
    // walk the candidate shapes (one iteration for single-candidate tiles: the adjoint itself tests the hit)
    int wd = 0;
    unsigned long long m = MULTI ? mk[0] : 1ull;
    while (true) {
      while (MULTI && m == 0 && ++wd < a.nwords) m = mk[wd];
      if (m == 0) break;
      const int k = MULTI ? wd * 64 + __builtin_ctzll(m) : k0;
      m &= m - 1;
      if (MULTI) {
        bool mine = false;
#pragma unroll
        for (int p = 0; p < P; ++p) mine |= (s_best_wave[p][lane] == k);
        if (!__ballot(mine)) continue;
      }
      ShapeP q;
      const float qword = shape_word(a, s_scene, t.b, k, lane);
      shape_head(qword, q);
      shape_material(qword, q);
      float acc[ACC_USED];
      // the empty asm pins each accumulator to one VGPR for the whole row loop: without it hipcc keeps
      // two copies of the block and shuffles 30 registers around the loop
#pragma unroll
      for (int n = 0; n < ACC_USED; ++n) { acc[n] = 0.f; asm volatile("" : "+v"(acc[n])); }
      bool anyhit = false;
      RRT_STAMP(4);
      if (!SH && SHADER == SHADER_PHONG && (q.powf & (SPEC ? FAST_SPEC : FAST_NOSPEC))) {
        // the hot path: a FAST_* Sphere (rrt_device.h: adjoint_sphere_phong).  Its most used uniforms --
        // the transform, the light and view directions -- as per-lane VGPR copies: on gfx950 a VALU
        // instruction with an SGPR source issues at about half the rate of one that reads only VGPRs
        // (tools/ubench/issue_probe2.hip: v_fmac 2.4 vs 4.3 cycles at two waves per SIMD)
        float Av[9], ov[3], Lv[3], lookv[3], shm1 = q.sh - 1.0f;
        load_shape_vgpr(a, s_scene, t.b, k, Av, ov);
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          Lv[n] = sc.L[n]; lookv[n] = sc.look[n];
          asm volatile("" : "+v"(Lv[n]), "+v"(lookv[n]));
        }
        asm volatile("" : "+v"(shm1));
        float gs_sum = 0.f;                          // sum of g_s over the rows: G_Lhat's g_s look_at part, applied once below
        auto fast_row = [&](int p, int s, float jcx, float jcy) {
          const bool on = MULTI ? (s_best_wave[p][lane] == k) : col_ok;
          if (on) {
            float d[3];
            row_ray<JC, TR>(a, lr, t.a0 + p, s, jcx, jcy, d);
            const float* ur = up_rows + p * UP_ROW + 3 * lane;
            const float up[3] = {ur[0], ur[1], ur[2]};
            const bool hit = adjoint_sphere_phong<float, SPEC>(Av, ov, q.colI, q.cmax, q.ks, q.kd, q.ka, shm1, q.kssh, q.powf,
                                                               Lv, lookv, sc.Llook, d, up, acc, &gs_sum);
            if (MULTI) anyhit |= hit;
          }
        };
        // single-candidate tiles do not track whether anything hit (a few per cent of them add zeros to the
        // accumulators: cheaper than the mask bookkeeping in every row).  (Rows unrolled: spills in the tile loop.)
        if (!MULTI) anyhit = true;
        // Wave priority: LOW in the row loop, raised again behind it.  The rows are the bulk of the VALU work and have
        // no latency to hide; everything around them is a chain of short latencies (queue pop, prefetch issue, LDS
        // reads of the shape, butterfly, LDS atomics).  With the arbiter serving those first, a wave is back in its
        // next row loop sooner and the SIMD sees more waves with VALU work at once (k_bwd 39.7 -> 37.8 us; the
        // opposite setting, rows first, costs 1.2 us)
        __builtin_amdgcn_s_setprio(0);
#pragma unroll 1
        for (int s = s_lo; s < s_hi; ++s) {
          float jcx, jcy;
          centre_jitter(v, s, jcx, jcy);
#pragma unroll 1
          for (int p = 0; p < t.nrows; ++p) fast_row(p, s, jcx, jcy);
        }
        __builtin_amdgcn_s_setprio(2);
        adjoint_sphere_phong_finish<float>(gs_sum, lookv, acc);
      } else if (SH || !JC) {
        // The generic adjoint inline.  The centre-jitter kernels call it as a function instead (bwd_generic_shape, below):
        // measured per kernel -- the call's register convention costs the hash / explicit-jitter kernels 4-12 us (C3,
        // C5), and the shadows extension walks the LDS scene through generic pointers, which this compiler cannot handle
        // inside a real function.
        shape_transform(qword, q);
        __builtin_amdgcn_s_setprio(0);               // (as above)
#pragma unroll 1
        for (int s = s_lo; s < s_hi; ++s) {
          float jcx, jcy;
          centre_jitter(v, s, jcx, jcy);
#pragma unroll 1
          for (int p = 0; p < t.nrows; ++p) {
            bool on = MULTI ? (s_best_wave[p][lane] == k) : col_ok;
            float d[3];
            if (SH) {
              // (extension) a shadowed hit is black: no gradient.  The distance of shape k along this ray, then the
              // wave-wide any-hit walk over the other spheres -- every lane takes part, `on` says who is concerned
              row_ray<JC, TR>(a, lr, t.a0 + p, s, jcx, jcy, d);
              float r[3], tt = 0.f;
              xform_dir<float>(q.A, d, r);
              bool hit;
              if (q.kind == KIND_SPHERE) { SphereHit<float> sp; hit = hit_sphere<float>(r, q.o, sp); tt = sp.t; }
              else hit = hit_square<float>(r, q.o, tt);
              on = on && hit;
              if (wave_in_shadow(a, s_scene, t.b, k, d, tt, on, lane)) on = false;
            }
            if (on) {
              if (!SH) row_ray<JC, TR>(a, lr, t.a0 + p, s, jcx, jcy, d);
              const float* ur = up_rows + p * UP_ROW + 3 * lane;
              const float up[3] = {ur[0], ur[1], ur[2]};
              anyhit |= adjoint_ray<float>(q.kind, q.A, q.o, q.colI, q.ks, q.kd, q.ka, q.sh, q.powf, SHADER, SPEC,
                                           v.inv_max_depth, sc.L, sc.look, d, up, acc);
            }
          }
        }
        __builtin_amdgcn_s_setprio(2);
      } else {
        typedef __attribute__((address_space(3))) const float* lcf;
        typedef __attribute__((address_space(3))) const int* lci;
        const GenericSums g = bwd_generic_shape<P, SHADER, SPEC, JC, TR, MULTI>(
            t.b, t.a0, t.nrows, k, s_lo, s_hi, lr.cl, lr.q, lr.col, col_ok ? 1 : 0, qword,
            (unsigned)(size_t)(lcf)up_rows, (unsigned)(size_t)(lci)&s_best_wave[0][0], (unsigned long long)kargs_here());
#pragma unroll
        for (int n = 0; n < ACC_USED; ++n) acc[n] = g.v[n];
        anyhit = g.anyhit != 0;
      }
      RRT_STAMP(5);
      if (!__ballot(anyhit)) continue;
      // 22 per-lane partial sums -> 22 totals, as a transposing butterfly: a swap + an add folds TWO
      // values across half the wave (v_permlane32_swap), then across rows (v_permlane16_swap); the six
      // survivors carry four values each (one per 16-lane row) and finish with four DPP adds -- 59
      // VALU ops instead of 22 full wave sums (176)
      float r6[6];
      wave_sums_22(acc, r6);
      RRT_STAMP(6);
      // every accumulator is linear in the upstream gradient: the 1/antialias_samples of the sample
      // average (scene.py:50) is applied to the six sums instead of to every gradient row
#pragma unroll
      for (int n = 0; n < 6; ++n) r6[n] *= v.inv_A;
      if (a.lds_scene) {
        // row j of r6[m] holds total m + {0,6,11,17}[j] in its last lane (r6[5]: rows 0 and 2 only)
        float* row = &s_acc[lds_rec(a, t.b, k) * ACC_STRIDE + (lane >> 4) * 6 - (lane >> 5)];
        if ((lane & 15) == 15) {
#pragma unroll
          for (int n = 0; n < 5; ++n) atomicAdd(row + n, r6[n]);
          if (!(lane & 16)) atomicAdd(row + 5, r6[5]);
        }
      } else {
        float mine_val = 0.f;
#pragma unroll
        for (int n = 0; n < ACC_USED; ++n) {
          const int mm = n < 6 ? n : (n < 11 ? n - 6 : (n < 16 ? n - 11 : (n == 16 ? 5 : n - 17)));
          const int j = n < 6 ? 0 : (n < 11 ? 1 : (n < 17 ? 2 : 3));
          const float bc = lane_bcast(r6[mm], 16 * j + 15);
          mine_val = (lane == n) ? bc : mine_val;
        }
        if (w.held && lane < ACC_USED)
          __hip_atomic_fetch_add(acc_copy(a) + ((size_t)w.held_b * a.S + w.held_k) * ACC_STRIDE + lane, w.held_val,
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w.held = true; w.held_k = k; w.held_b = t.b; w.held_val = mine_val;
      }
      RRT_STAMP(7);
      RRT_STAMP_VAL(13, blockIdx.x);
    }
  }
  RRT_STAMP(8);
  RRT_STAMP_RT(15);
  w.e_cur = w.e_nxt;
  if (SP) w.part_cur = w.part_nxt;
  w.has_cur = has_nxt;
  w.buf ^= 1;
}

// Persistent over the tiles that have candidates (tiles without never touch
// the upstream gradient), pulled from the workgroup's LDS queue.  The next
// tile's gradient rows are requested before the current tile is processed, so
// HBM latency hides behind the adjoint.
template <int P, int SHADER, int SPEC, int JC, int TR, int SH, int SP = 0>
__global__ __launch_bounds__(BLOCK, RRT_BWD_WAVES_PER_SIMD) void k_bwd(KArgs a) {
  __shared__ float s_scene[LDS_RECS * PS_STRIDE];
  __shared__ WgQueue s_q;
  // wave-private staging: the upstream gradient rows of the current and of the next tile (LDS-DMA
  // double buffer), and the winner id per (row, lane).  The winners live in the queue's scan tables,
  // which are dead once a batch has been dealt (152 KB of the CU's 160 KB at P = 4).
  __shared__ float s_up[WAVES][2][P][UP_ROW];
  static_assert(sizeof(s_q.ex) + sizeof(s_q.cnt01) >= sizeof(int) * WAVES * P * 64 &&
                offsetof(WgQueue, cnt01) == offsetof(WgQueue, ex) + sizeof(s_q.ex), "s_best aliases ex + cnt01");
  int (*s_best)[P][64] = reinterpret_cast<int (*)[P][64]>(&s_q.ex[0][0]);
  // workgroup-level gradient accumulators: a wave's 22 sums go to LDS (ds_add_f32) and the
  // workgroup adds each touched row to HBM once, at the end.  A same-address global atomic
  // costs ~12 ns and every tile of a shape adds to the same 22 words: one big sphere
  // (57 k tiles) made k_bwd 758 us of pure atomic serialisation.
  __shared__ float s_acc[LDS_RECS * ACC_STRIDE];
  const ViewK& v = a.v;
  int lane;
  global_wave<WAVES>(lane);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // chunk of list segments per batch: one per thread, and no more than Q_CAP entries for this workgroup
  const unsigned chunk = (unsigned)a.chunk;
  // (in flight under the staging: the saved queue batch, or the segment sizes to deal one)
  uint2 cw = make_uint2(0u, 0u);
  unsigned saved_cnt = 0;
  unsigned long long saved_ent = 0;
  if (a.use_saved) {
    saved_cnt = as_const_u32(a.wgq_cnt)[blockIdx.x];
    // (the whole slot is allocated: read without waiting for the count -- one memory round trip less at the kernel's
    // start, k_bwd -0.3 us; entries beyond the count are never used)
    saved_ent = a.wgq[(size_t)blockIdx.x * Q_CAP + threadIdx.x];
  } else if (threadIdx.x < min(chunk, (unsigned)a.n_bins)) {
    cw = reinterpret_cast<const uint2*>(a.blk_counts)[threadIdx.x];
  }
  if (a.lds_scene == 1)
    for (int i = threadIdx.x; i < a.B * a.S * ACC_STRIDE; i += blockDim.x) s_acc[i] = 0.f;
  stage_scene(a, s_scene);          // ends with the workgroup barrier
  BwdWave w;
  w.e_cur = w.e_nxt = 0; w.has_cur = false; w.buf = 0; w.b_cur = -1; w.part_cur = w.part_nxt = 0;
  w.sc = {{0.f, 0.f, 1.f}, {0.f, 0.f, 1.f}, 1.f};
  w.held = false; w.held_k = w.held_b = 0; w.held_val = 0.f;
  float* const s_up_wave = &s_up[wave][0][0][0];
  int (*const s_best_wave)[64] = s_best[wave];

  QueueState qs;
  qs.next = blockIdx.x;
  // lds_scene = 2 (a batch of many small scenes): the workgroup takes whole list segments, stages the segment's
  // scene, accumulates the segment's gradient in LDS and adds it to HBM when the segment is done
  const bool local = a.lds_scene == 2;
  for (unsigned bin0 = local ? blockIdx.x : 0u; bin0 < (unsigned)a.n_bins; bin0 += local ? gridDim.x : chunk) {
    unsigned cnt;
    if (local) {
      if (bin0 != blockIdx.x) {
        __syncthreads();                     // every wave is done adding to the LDS rows of the previous segment
        const int bp = (int)((bin0 - gridDim.x) / (unsigned)a.bins_per_scene);
        for (int i = threadIdx.x; i < a.S * ACC_STRIDE; i += blockDim.x) {
          const float x = s_acc[i];
          if ((i & (ACC_STRIDE - 1)) < ACC_USED && x != 0.f)
            __hip_atomic_fetch_add(acc_copy(a) + (size_t)bp * a.S * ACC_STRIDE + i, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
      }
      const uint2 c = reinterpret_cast<const uint2*>(a.blk_counts)[bin0];
      cnt = (c.x & 0xffffu) + (c.x >> 16) + (c.y & 0xffffu);
      stage_one_scene(a, s_scene, (int)(bin0 / (unsigned)a.bins_per_scene));
      for (int i = threadIdx.x; i < a.S * ACC_STRIDE; i += blockDim.x) s_acc[i] = 0.f;
      if (threadIdx.x < cnt) s_q.ent[threadIdx.x] = a.heavy[(size_t)bin0 * SEG + threadIdx.x];
      if (threadIdx.x == 0) s_q.cursor = 0;
      __syncthreads();
    } else {
      const unsigned nb = min(chunk, (unsigned)a.n_bins - bin0);
      const size_t slot = (size_t)(bin0 / chunk) * gridDim.x + blockIdx.x;
      if (bin0) {
        __syncthreads();                     // previous batch fully consumed
        if (a.use_saved) {
          saved_cnt = as_const_u32(a.wgq_cnt)[slot];
          saved_ent = a.wgq[slot * Q_CAP + threadIdx.x];
        } else {
          cw = threadIdx.x < nb ? reinterpret_cast<const uint2*>(a.blk_counts)[bin0 + threadIdx.x] : make_uint2(0u, 0u);
        }
      }
      if (a.use_saved) {
        // the forward pass on this setup dealt exactly this share: one contiguous read
        cnt = saved_cnt;
        if (threadIdx.x < cnt) s_q.ent[threadIdx.x] = saved_ent;
        if (threadIdx.x == 0) s_q.cursor = 0;
        __syncthreads();
      } else {
        cnt = queue_fill<false>(s_q, nullptr, a.heavy, nullptr, cw, bin0, nb, qs, lane, wave, NoHook(),
                                a.wgq + slot * Q_CAP, a.wgq_cnt + slot);
      }
    }
    w.has_cur = queue_pop_part<SP>(s_q, cnt, lane, w.e_cur, w.part_cur);
    w.buf = 0;
    if (w.has_cur) {
      int k0; bool multi; long long tile;
      const Tile t = decode_entry(a, w.e_cur, k0, multi, tile);
      prefetch_grad_rows<P>(a, t, lane, s_up_wave);
    }
    // the queue is ordered by cost class: first the tiles with several candidates, then the others
    while (w.has_cur && entry_is_multi(w.e_cur))
      bwd_tile<P, SHADER, SPEC, JC, TR, true, SH, SP>(a, w, s_q, cnt, s_scene, s_acc, s_up_wave, s_best_wave, lane);
    // (queue_fill stages a workgroup's share in list order -- class 0, 1, 2 -- and the saved batches keep
    // that order: no entry with several candidates follows one with a single candidate)
    while (w.has_cur)
      bwd_tile<P, SHADER, SPEC, JC, TR, false, SH, SP>(a, w, s_q, cnt, s_scene, s_acc, s_up_wave, s_best_wave, lane);
  }
  if (w.held && lane < ACC_USED)
    __hip_atomic_fetch_add(acc_copy(a) + ((size_t)w.held_b * a.S + w.held_k) * ACC_STRIDE + lane, w.held_val, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  if (a.lds_scene) {
    __syncthreads();                 // every wave of the workgroup is done adding
    // lds_scene = 2: the rows of the LAST segment this workgroup worked on (if it had any)
    const unsigned last = local && blockIdx.x < (unsigned)a.n_bins
                              ? blockIdx.x + ((unsigned)a.n_bins - 1u - blockIdx.x) / gridDim.x * gridDim.x : 0u;
    const int n = local ? (blockIdx.x < (unsigned)a.n_bins ? a.S * ACC_STRIDE : 0) : a.B * a.S * ACC_STRIDE;
    float* dst = acc_copy(a) + (local ? (size_t)(last / (unsigned)a.bins_per_scene) * a.S * ACC_STRIDE : 0);
    // (every workgroup starts at a different shape row: they finish together and would otherwise all queue on row 0)
    const int rot = n ? (int)((blockIdx.x * 5u * ACC_STRIDE) % (unsigned)n) : 0;
    for (int i0 = threadIdx.x; i0 < n; i0 += blockDim.x) {
      int i = i0 + rot;
      if (i >= n) i -= n;
      const float x = s_acc[i];
      if ((i & (ACC_STRIDE - 1)) < ACC_USED && x != 0.f)
        __hip_atomic_fetch_add(dst + i, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// --------------------------------------------------------------- finalize --
// One block; the work is tiny (B*S rows of 32 floats) and the kernel is pure latency.  Per batch of 256
// rows: all threads sum the R copies of the rows (k_bwd's workgroups add to copy blockIdx mod R), float4
// by float4 with the R loads of a thread in flight together, leave the sums in LDS and the copies zeroed
// for the next backward; then a thread per row derives all three gradients from its row -- stored directly
// where every scene has its own transforms and materials (or there is one scene), summed over the scenes
// through LDS (ds_add_f32) for shared (stride-0) parameters -- and the light terms are summed with DPP
// wave sums and one barrier.  Scenes of more than FIN_S shapes with shared parameters keep serial loops
// (they run with one copy).
constexpr int FIN_S = 256, FIN_ROWS = 256;
template <int R>
__global__ __launch_bounds__(256) void k_finalize(int S, int B, float* acc, const float* hdr, const float* mat,
                                                  long long mat_bs, const float* light, long long w2o_bs,
                                                  float* g_w2o, float* g_mat, float* g_light) {
  __shared__ float s_red[256][6];
  __shared__ float4 s_rows[FIN_ROWS * (ACC_STRIDE / 4)];
  __shared__ float s_gw[FIN_S * 12];
  __shared__ float s_gm[FIN_S * 7];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Bw = w2o_bs ? B : 1, Bm = mat_bs ? B : 1;
  const int n_rows = B * S;
  const bool own = B == 1 || (w2o_bs && mat_bs);           // nothing is summed over the scenes
  if (own || S <= FIN_S) {
    if (!own) {
      if (g_w2o && !w2o_bs)
        for (int i = tid; i < S * 12; i += blockDim.x) s_gw[i] = 0.f;
      if (g_mat && !mat_bs)
        for (int i = tid; i < S * 7; i += blockDim.x) s_gm[i] = 0.f;
    }
    float gl[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const size_t copy4 = (size_t)n_rows * (ACC_STRIDE / 4);             // float4s per copy
    for (int base = 0; base < n_rows; base += FIN_ROWS) {
      const int nb = min(FIN_ROWS, n_rows - base);
      if (base) __syncthreads();                                        // the previous batch's rows are consumed
      // what this thread's row will need besides its sums -- camera rows, light intensity, the shape's colour -- is
      // requested here, together with the sums: behind the barrier it would be a second memory round trip
      float hc[HDR_STRIDE], li[6], col[7];
      if (tid < nb) {
        const int idx = base + tid, b = idx / S, k = idx - b * S;
#pragma unroll
        for (int n = 0; n < 12; ++n) hc[HDR_CAM + n] = hdr[(size_t)b * HDR_STRIDE + HDR_CAM + n];     // camera 9 + origin 3
#pragma unroll
        for (int n = 0; n < 3; ++n) { li[3 + n] = light[3 + n]; col[n] = mat[b * mat_bs + 7 * k + n]; }
      }
      float4* src = reinterpret_cast<float4*>(acc) + (size_t)base * (ACC_STRIDE / 4);
      for (int c = tid; c < nb * (ACC_STRIDE / 4); c += blockDim.x) {
        float4 x[R];                                                    // (R is a template parameter: the loads leave together)
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = src[(size_t)r * copy4 + c];
#pragma unroll
        for (int r = 0; r < R; ++r) src[(size_t)r * copy4 + c] = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 sum = x[0];
#pragma unroll
        for (int r = 1; r < R; ++r) { sum.x += x[r].x; sum.y += x[r].y; sum.z += x[r].z; sum.w += x[r].w; }
        s_rows[c] = sum;
      }
      __syncthreads();
      if (tid < nb) {
        const int idx = base + tid, b = idx / S, k = idx - b * S;
        float row[ACC_STRIDE];
#pragma unroll
        for (int q = 0; q < ACC_STRIDE / 4; ++q) {
          const float4 x = s_rows[tid * (ACC_STRIDE / 4) + q];
          row[4 * q] = x.x; row[4 * q + 1] = x.y; row[4 * q + 2] = x.z; row[4 * q + 3] = x.w;
        }
        if (g_w2o) {
          float gw[12];
          for (int n = 0; n < 12; ++n) gw[n] = 0.f;
          finalize_w2o<float>(row, hc, gw);
          if (own || w2o_bs) { for (int n = 0; n < 12; ++n) g_w2o[(size_t)idx * 12 + n] = gw[n]; }
          else { for (int n = 0; n < 12; ++n) atomicAdd(&s_gw[k * 12 + n], gw[n]); }
        }
        if (g_mat) {
          float gm[7];
          for (int n = 0; n < 7; ++n) gm[n] = 0.f;
          finalize_material<float>(row, li, gm);
          if (own || mat_bs) { for (int n = 0; n < 7; ++n) g_mat[(size_t)idx * 7 + n] = gm[n]; }
          else { for (int n = 0; n < 7; ++n) atomicAdd(&s_gm[k * 7 + n], gm[n]); }
        }
        if (g_light) finalize_light_accum<float>(row, col, gl);
      }
    }
    if (g_light) {
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        const float tot = wave_sum_to_lane63(gl[n]);
        if (lane == 63) s_red[wave][n] = tot;
      }
    }
    __syncthreads();
    if (!own) {
      if (g_w2o && !w2o_bs)
        for (int i = tid; i < S * 12; i += blockDim.x) g_w2o[i] = s_gw[i];
      if (g_mat && !mat_bs)
        for (int i = tid; i < S * 7; i += blockDim.x) g_mat[i] = s_gm[i];
    }
    if (g_light && tid == 0) {
      float g[6];
      for (int n = 0; n < 6; ++n) g[n] = s_red[0][n] + s_red[1][n] + s_red[2][n] + s_red[3][n];
      finalize_light_dir<float>(hdr, g);            // the light is shared by all scenes
      for (int n = 0; n < 6; ++n) g_light[n] = g[n];
    }
    return;
  }
  // shared parameters and more than FIN_S shapes: serial loops over the scenes (one copy of the rows: R == 1)
  if (g_w2o) {
    for (int idx = tid; idx < Bw * S; idx += blockDim.x) {
      int bw = idx / S, k = idx - bw * S;
      float gw[12];
      for (int n = 0; n < 12; ++n) gw[n] = 0.f;
      int blo = w2o_bs ? bw : 0, bhi = w2o_bs ? bw + 1 : B;
      for (int b = blo; b < bhi; ++b)
        finalize_w2o<float>(acc + ((size_t)b * S + k) * ACC_STRIDE, hdr + (size_t)b * HDR_STRIDE, gw);
      for (int n = 0; n < 12; ++n) g_w2o[(size_t)idx * 12 + n] = gw[n];
    }
  }
  if (g_mat) {
    for (int idx = tid; idx < Bm * S; idx += blockDim.x) {
      int bm = idx / S, k = idx - bm * S;
      float gm[7];
      for (int n = 0; n < 7; ++n) gm[n] = 0.f;
      int blo = mat_bs ? bm : 0, bhi = mat_bs ? bm + 1 : B;
      for (int b = blo; b < bhi; ++b) finalize_material<float>(acc + ((size_t)b * S + k) * ACC_STRIDE, light, gm);
      for (int n = 0; n < 7; ++n) g_mat[(size_t)idx * 7 + n] = gm[n];
    }
  }
  if (g_light) {
    float gl[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int idx = tid; idx < B * S; idx += blockDim.x) {
      int b = idx / S, k = idx - b * S;
      finalize_light_accum<float>(acc + (size_t)idx * ACC_STRIDE, mat + b * mat_bs + 7 * k, gl);
    }
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const float tot = wave_sum_to_lane63(gl[n]);
      if (lane == 63) s_red[wave][n] = tot;
    }
    __syncthreads();
    if (tid == 0) {
      float g[6];
      for (int n = 0; n < 6; ++n) g[n] = s_red[0][n] + s_red[1][n] + s_red[2][n] + s_red[3][n];
      finalize_light_dir<float>(hdr, g);            // the light is shared by all scenes
      for (int n = 0; n < 6; ++n) g_light[n] = g[n];
    }
  }
  __syncthreads();                                   // every read of acc is done
  for (int idx = tid; idx < B * S * (ACC_STRIDE / 4); idx += blockDim.x)
    reinterpret_cast<float4*>(acc)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// The single-copy form (R = 1: batches of many small scenes, scenes too big for the LDS accumulators): every row is read
// ONCE straight from memory (float4 loads, all in flight together) -- no LDS hop, all row batches' loads in flight at once
// (through the staged form above the 512 rows of the 256-scene orbit batch took 12.8 instead of 8.2 us).
// One block; the work is tiny (B*S rows of 32 floats) and the kernel is pure latency, so
// the common case -- every scene has its own transforms and materials, or there is one scene
// -- reads each accumulator row ONCE (float4 loads, all in flight together), derives all three
// gradients from it, re-zeroes it, and sums the light terms with DPP wave sums and one
// barrier.  Shared (stride-0) parameters take the general path below.  Leaves the accumulator
// rows zeroed for the next backward.
__global__ __launch_bounds__(256) void k_finalize_one(int S, int B, float* acc, const float* hdr, const float* mat,
                                                  long long mat_bs, const float* light, long long w2o_bs,
                                                  float* g_w2o, float* g_mat, float* g_light) {
  __shared__ float s_red[256][6];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Bw = w2o_bs ? B : 1, Bm = mat_bs ? B : 1;
  if (B == 1 || (w2o_bs && mat_bs)) {
    float gl[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int idx = tid; idx < B * S; idx += blockDim.x) {
      const int b = idx / S, k = idx - b * S;
      float row[ACC_STRIDE];
      float4* src = reinterpret_cast<float4*>(acc + (size_t)idx * ACC_STRIDE);
#pragma unroll
      for (int q = 0; q < ACC_STRIDE / 4; ++q) {
        const float4 x = src[q];
        row[4 * q] = x.x; row[4 * q + 1] = x.y; row[4 * q + 2] = x.z; row[4 * q + 3] = x.w;
      }
#pragma unroll
      for (int q = 0; q < ACC_STRIDE / 4; ++q) src[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g_w2o) {
        float gw[12];
        for (int n = 0; n < 12; ++n) gw[n] = 0.f;
        finalize_w2o<float>(row, hdr + (size_t)b * HDR_STRIDE, gw);
        for (int n = 0; n < 12; ++n) g_w2o[(size_t)idx * 12 + n] = gw[n];
      }
      if (g_mat) {
        float gm[7];
        for (int n = 0; n < 7; ++n) gm[n] = 0.f;
        finalize_material<float>(row, light, gm);
        for (int n = 0; n < 7; ++n) g_mat[(size_t)idx * 7 + n] = gm[n];
      }
      if (g_light) finalize_light_accum<float>(row, mat + b * mat_bs + 7 * k, gl);
    }
    if (g_light) {
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        const float tot = wave_sum_to_lane63(gl[n]);
        if (lane == 63) s_red[wave][n] = tot;
      }
      __syncthreads();
      if (tid == 0) {
        float g[6];
        for (int n = 0; n < 6; ++n) g[n] = s_red[0][n] + s_red[1][n] + s_red[2][n] + s_red[3][n];
        finalize_light_dir<float>(hdr, g);            // the light is shared by all scenes
        for (int n = 0; n < 6; ++n) g_light[n] = g[n];
      }
    }
    return;
  }
  // Shared (batch stride 0) transforms or materials: their gradient is the sum over the scenes.  Every
  // thread chains one (scene, shape) row and adds it to LDS sums (a serial loop over the scenes per shape
  // made the 256-scene orbit batch spend 30 us here); beyond FIN_S shapes the serial loops remain.
  __shared__ float s_gw[FIN_S * 12];
  __shared__ float s_gm[FIN_S * 7];
  if (S <= FIN_S) {
    if (g_w2o && !w2o_bs)
      for (int i = tid; i < S * 12; i += blockDim.x) s_gw[i] = 0.f;
    if (g_mat && !mat_bs)
      for (int i = tid; i < S * 7; i += blockDim.x) s_gm[i] = 0.f;
    __syncthreads();
    float gl[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int idx = tid; idx < B * S; idx += blockDim.x) {
      const int b = idx / S, k = idx - b * S;
      float row[ACC_STRIDE];
      const float4* src = reinterpret_cast<const float4*>(acc + (size_t)idx * ACC_STRIDE);
#pragma unroll
      for (int q = 0; q < ACC_STRIDE / 4; ++q) {
        const float4 x = src[q];
        row[4 * q] = x.x; row[4 * q + 1] = x.y; row[4 * q + 2] = x.z; row[4 * q + 3] = x.w;
      }
      if (g_w2o) {
        float gw[12];
        for (int n = 0; n < 12; ++n) gw[n] = 0.f;
        finalize_w2o<float>(row, hdr + (size_t)b * HDR_STRIDE, gw);
        if (w2o_bs) { for (int n = 0; n < 12; ++n) g_w2o[(size_t)idx * 12 + n] = gw[n]; }
        else { for (int n = 0; n < 12; ++n) atomicAdd(&s_gw[k * 12 + n], gw[n]); }
      }
      if (g_mat) {
        float gm[7];
        for (int n = 0; n < 7; ++n) gm[n] = 0.f;
        finalize_material<float>(row, light, gm);
        if (mat_bs) { for (int n = 0; n < 7; ++n) g_mat[(size_t)idx * 7 + n] = gm[n]; }
        else { for (int n = 0; n < 7; ++n) atomicAdd(&s_gm[k * 7 + n], gm[n]); }
      }
      if (g_light) finalize_light_accum<float>(row, mat + b * mat_bs + 7 * k, gl);
    }
    if (g_light) {
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        const float tot = wave_sum_to_lane63(gl[n]);
        if (lane == 63) s_red[wave][n] = tot;
      }
    }
    __syncthreads();
    if (g_w2o && !w2o_bs)
      for (int i = tid; i < S * 12; i += blockDim.x) g_w2o[i] = s_gw[i];
    if (g_mat && !mat_bs)
      for (int i = tid; i < S * 7; i += blockDim.x) g_mat[i] = s_gm[i];
    if (g_light && tid == 0) {
      float g[6];
      for (int n = 0; n < 6; ++n) g[n] = s_red[0][n] + s_red[1][n] + s_red[2][n] + s_red[3][n];
      finalize_light_dir<float>(hdr, g);            // the light is shared by all scenes
      for (int n = 0; n < 6; ++n) g_light[n] = g[n];
    }
  } else {
    if (g_w2o) {
      for (int idx = tid; idx < Bw * S; idx += blockDim.x) {
        int bw = idx / S, k = idx - bw * S;
        float gw[12];
        for (int n = 0; n < 12; ++n) gw[n] = 0.f;
        int blo = w2o_bs ? bw : 0, bhi = w2o_bs ? bw + 1 : B;
        for (int b = blo; b < bhi; ++b)
          finalize_w2o<float>(acc + ((size_t)b * S + k) * ACC_STRIDE, hdr + (size_t)b * HDR_STRIDE, gw);
        for (int n = 0; n < 12; ++n) g_w2o[(size_t)idx * 12 + n] = gw[n];
      }
    }
    if (g_mat) {
      for (int idx = tid; idx < Bm * S; idx += blockDim.x) {
        int bm = idx / S, k = idx - bm * S;
        float gm[7];
        for (int n = 0; n < 7; ++n) gm[n] = 0.f;
        int blo = mat_bs ? bm : 0, bhi = mat_bs ? bm + 1 : B;
        for (int b = blo; b < bhi; ++b) finalize_material<float>(acc + ((size_t)b * S + k) * ACC_STRIDE, light, gm);
        for (int n = 0; n < 7; ++n) g_mat[(size_t)idx * 7 + n] = gm[n];
      }
    }
    if (g_light) {
      float gl[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int idx = tid; idx < B * S; idx += blockDim.x) {
        int b = idx / S, k = idx - b * S;
        finalize_light_accum<float>(acc + (size_t)idx * ACC_STRIDE, mat + b * mat_bs + 7 * k, gl);
      }
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        const float tot = wave_sum_to_lane63(gl[n]);
        if (lane == 63) s_red[wave][n] = tot;
      }
      __syncthreads();
      if (tid == 0) {
        float g[6];
        for (int n = 0; n < 6; ++n) g[n] = s_red[0][n] + s_red[1][n] + s_red[2][n] + s_red[3][n];
        finalize_light_dir<float>(hdr, g);            // the light is shared by all scenes
        for (int n = 0; n < 6; ++n) g_light[n] = g[n];
      }
    }
  }
  __syncthreads();                                   // every read of acc is done
  for (int idx = tid; idx < B * S * (ACC_STRIDE / 4); idx += blockDim.x)
    reinterpret_cast<float4*>(acc)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------- host --
thread_local int g_last_hip_error = 0;

int hip_fail(hipError_t e) {
  g_last_hip_error = (int)e;
  return RRT_ERR_HIP;
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Geometry {
  int P, ncols, nwords, tiles_x, tiles_y;
  long long n_tiles, n_bins;
  size_t ps_off, hdr_off, acc_off, counts_off, mask_off, heavy_off, empty_off, wgq_off, wgq_cnt_off, total;
  unsigned blocks_fwd, blocks_bwd;   // persistent grids of the render kernels
  int lds_mode;                      // KArgs::lds_scene
  int chunk;                         // list segments per queue batch
  int acc_repl;                      // copies of the gradient-accumulator rows (KArgs::acc_repl)
};

// RRT_TUNE_* environment variables (tuning experiments and tests only), read ONCE per process -- not on
// every call of the product path.  Tests that change them call rrt_debug_reload_tuning().
struct Tune {
  int P = 0;              // RRT_TUNE_P=1|2|4: rows per wave tile
  int fwd_wgs = 0, bwd_wgs = 0, chunk = 0;
  int repl = 4;                  // RRT_TUNE_REPL=1|2|4|8: copies of the gradient-accumulator rows (KArgs::acc_repl)
  int ztail = 20;                // RRT_TUNE_ZTAIL=0..90: per cent of the zero-filled tiles kept for the end of the render loop
                                 // (sweep 0 / 15 / 30 / 50 / 70: 71.9 / 71.5 / 71.6 / 72.3 / 73.0 us per step, three rounds)
  int zw_even = 6, zw_odd = 5;   // RRT_TUNE_ZW=even:odd  shares of k_fwd's zero-filled tiles by blockIdx parity (XcdDeal;
                                 // sweep 1:1 ... 4:3, profiles/r03_xcd_store_asymmetry.txt: flat between 7:6 and 5:4)
  int split = -1;                // RRT_TUNE_SPLIT=0|2|4: waves that share a tile's samples in k_bwd (KArgs::split_log2); -1 = by size
  bool no_lds = false, no_saved = false;
};
Tune g_tune;
std::once_flag g_tune_once;
void read_tune(Tune& x) {
  auto num = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; };
  const int p = num("RRT_TUNE_P");
  x.P = (p == 1 || p == 2 || p == 4) ? p : 0;
  x.fwd_wgs = num("RRT_TUNE_FWD_WGS");
  x.bwd_wgs = num("RRT_TUNE_BWD_WGS");
  x.chunk = num("RRT_TUNE_CHUNK");
  if (getenv("RRT_TUNE_ZTAIL")) { const int z = num("RRT_TUNE_ZTAIL"); x.ztail = z < 0 ? 0 : (z > 90 ? 90 : z); }
  if (getenv("RRT_TUNE_REPL")) { const int r = num("RRT_TUNE_REPL"); x.repl = r >= 8 ? 8 : (r >= 4 ? 4 : (r >= 2 ? 2 : 1)); }
  if (const char* zw = getenv("RRT_TUNE_ZW")) {
    int we = 0, wo = 0;
    if (sscanf(zw, "%d:%d", &we, &wo) == 2 && wo >= 1 && we >= wo && we <= 64) { x.zw_even = we; x.zw_odd = wo; }
  }
  x.split = -1;
  if (getenv("RRT_TUNE_SPLIT")) { const int sp = num("RRT_TUNE_SPLIT"); x.split = sp >= 4 ? 4 : (sp >= 2 ? 2 : 0); }
  x.no_lds = getenv("RRT_TUNE_NOLDS") != nullptr;
  x.no_saved = getenv("RRT_TUNE_NOSAVED") != nullptr;
}
const Tune& tune() {
  std::call_once(g_tune_once, [] { read_tune(g_tune); });
  return g_tune;
}
int tuned_P() { return tune().P; }

// Persistent launches: enough workgroups to fill every CU at the kernels' register-limited occupancy,
// never more than there are tiles.  The CU count is looked up once per device.
int device_cus() {
  static std::mutex mu;
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  std::lock_guard<std::mutex> lock(mu);
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

unsigned persistent_blocks(long long n_tiles, int wgs_per_cu) {
  long long want = (long long)device_cus() * wgs_per_cu;
  long long need = (n_tiles + WAVES - 1) / WAVES;
  if (need < 1) need = 1;
  return (unsigned)(want < need ? want : need);
}


Geometry geometry_of(const rrt_scene* s, const rrt_view* v) {
  Geometry g;
  g.ncols = v->transpose ? v->y_dims : v->x_dims;
  g.nwords = (s->S + 63) / 64;
  g.tiles_x = (g.ncols + TILE_W - 1) / TILE_W;
  long long t1 = (long long)g.tiles_x * v->rows * s->B;     // tiles at P = 1
  g.P = t1 >= 4 * 8192 ? 4 : (t1 >= 2 * 8192 ? 2 : 1);
  if (tuned_P()) g.P = tuned_P();
  g.tiles_y = (v->rows + g.P - 1) / g.P;
  g.n_tiles = (long long)g.tiles_x * g.tiles_y * s->B;
  size_t n_shapes = (size_t)s->B * (size_t)s->S;
  g.ps_off = 0;
  g.hdr_off = g.ps_off + n_shapes * PS_STRIDE * sizeof(float);
  g.acc_off = g.hdr_off + (size_t)s->B * HDR_STRIDE * sizeof(float);
  g.n_bins = (long long)((g.tiles_x + BIN_RX - 1) / BIN_RX) * ((g.tiles_y + BIN_RY - 1) / BIN_RY) * s->B;
  // copies of the accumulator rows: where ALL of k_bwd's workgroups add to every row at the end of the kernel (LDS
  // accumulators for the whole batch, lds_mode 1 -- decided below from the same numbers) and k_finalize sums rows through LDS
  g.acc_repl = (!tune().no_lds && (long long)s->B * s->S <= LDS_RECS && s->S <= 256) ? tune().repl : 1;
  g.counts_off = align_up(g.acc_off + (size_t)g.acc_repl * n_shapes * ACC_STRIDE * sizeof(float), 16);
  g.mask_off = align_up(g.counts_off + (size_t)g.n_bins * 2 * sizeof(unsigned), 16);
  g.heavy_off = align_up(g.mask_off + (size_t)g.n_tiles * g.nwords * sizeof(unsigned long long), 16);
  g.empty_off = g.heavy_off + (size_t)g.n_bins * SEG * sizeof(unsigned long long);
  // saved queue batches: one Q_CAP-entry slot per (batch of list segments, workgroup)
  g.blocks_fwd = persistent_blocks(g.n_tiles, tune().fwd_wgs > 0 ? tune().fwd_wgs : 1);
  g.blocks_bwd = persistent_blocks(g.n_tiles, tune().bwd_wgs > 0 ? tune().bwd_wgs : 1);
  // where the packed scene and the gradient accumulators live while the render kernels run (KArgs::lds_scene)
  const long long recs = (long long)s->B * s->S;
  g.lds_mode = tune().no_lds ? 0 : (recs <= LDS_RECS ? 1 : (s->S <= LDS_RECS ? 2 : 0));
  if (g.lds_mode == 2) {             // a workgroup per list segment at a time
    if ((long long)g.blocks_fwd > g.n_bins) g.blocks_fwd = (unsigned)g.n_bins;
    if ((long long)g.blocks_bwd > g.n_bins) g.blocks_bwd = (unsigned)g.n_bins;
  }
  const unsigned blocks = g.blocks_fwd > g.blocks_bwd ? g.blocks_fwd : g.blocks_bwd;
  const unsigned fewest = g.blocks_fwd < g.blocks_bwd ? g.blocks_fwd : g.blocks_bwd;
  long long chunk = (long long)Q_CAP * fewest / SEG;
  chunk = chunk < 1 ? 1 : (chunk > BLOCK ? BLOCK : chunk);
  const int forced = tune().chunk;                             // tests: several batches at small sizes
  if (forced > 0 && forced < chunk) chunk = forced;
  g.chunk = (int)chunk;
  const size_t slots = (size_t)((g.n_bins + chunk - 1) / chunk) * blocks;
  g.wgq_off = align_up(g.empty_off + (size_t)g.n_bins * SEG * sizeof(unsigned long long), 16);
  g.wgq_cnt_off = g.wgq_off + slots * Q_CAP * sizeof(unsigned long long);
  g.total = align_up(g.wgq_cnt_off + slots * sizeof(unsigned), 16);
  return g;
}

int validate(const rrt_scene* s, const rrt_view* v) {
  if (!s || !v) return RRT_ERR_NULL;
  if (s->S < 0 || s->B < 1) return RRT_ERR_SHAPE;
  if (s->S > RRT_MAX_SHAPES) return RRT_ERR_TOO_MANY;
  if (s->S > 0 && (!s->w2o || !s->material)) return RRT_ERR_NULL;
  if (!s->light || !v->cam) return RRT_ERR_NULL;
  if (s->w2o_bstride != 0 && s->w2o_bstride != (int64_t)s->S * 12) return RRT_ERR_SHAPE;
  if (s->material_bstride != 0 && s->material_bstride != (int64_t)s->S * 7) return RRT_ERR_SHAPE;
  if (v->cam_bstride != 0 && v->cam_bstride != 16) return RRT_ERR_SHAPE;
  if (v->x_dims < 1 || v->y_dims < 1 || v->antialias_samples < 1) return RRT_ERR_SHAPE;
  int full = v->transpose ? v->x_dims : v->y_dims;
  if (v->row0 < 0 || v->rows < 0 || v->row0 + v->rows > full) return RRT_ERR_SHAPE;
  if (v->shader_id != RRT_SHADER_PHONG && v->shader_id != RRT_SHADER_DEPTH) return RRT_ERR_ENUM;
  if (v->jitter_mode < RRT_JITTER_CENTER || v->jitter_mode > RRT_JITTER_HASH) return RRT_ERR_ENUM;
  if (v->jitter_mode == RRT_JITTER_EXPLICIT && !v->jitter) return RRT_ERR_NULL;
  if (v->shader_id == RRT_SHADER_DEPTH && !(v->max_depth != 0.f)) return RRT_ERR_SHAPE;
  long long tiles1 = (long long)((v->transpose ? v->y_dims : v->x_dims) + TILE_W - 1) / TILE_W * v->rows * s->B;
  if (tiles1 > 0x7fffffffLL) return RRT_ERR_SHAPE;
  // packed tile-list entries: 16-bit tile column, 20-bit tile row, 16-bit scene
  if (((v->transpose ? v->y_dims : v->x_dims) + TILE_W - 1) / TILE_W > (1 << ENT_TX_BITS) ||
      v->rows > (1 << ENT_TY_BITS) || s->B > (1 << ENT_B_BITS))
    return RRT_ERR_SHAPE;
  return RRT_OK;
}

KArgs make_args(const rrt_scene* s, const rrt_view* v, void* workspace, const Geometry& g) {
  KArgs a;
  ViewK& k = a.v;
  k.x_dims = v->x_dims; k.y_dims = v->y_dims; k.A = v->antialias_samples;
  k.row0 = v->row0; k.rows = v->rows;
  k.ncols = g.ncols;
  k.transpose = v->transpose ? 1 : 0;
  k.jitter_mode = v->jitter_mode; k.seed = v->seed;
  k.shader = v->shader_id; k.specular = v->specular ? 1 : 0;
  k.flags = v->flags;
  // np.linspace(0.5,-0.5,y_dims) / np.linspace(-0.5,0.5,x_dims) steps (scene.py:66-67)
  k.ustep = v->y_dims > 1 ? (float)(-1.0 / (double)(v->y_dims - 1)) : 0.f;
  k.vstep = v->x_dims > 1 ? (float)(1.0 / (double)(v->x_dims - 1)) : 0.f;
  k.inv_xd = (float)(1.0 / (double)v->x_dims);
  k.inv_yd = (float)(1.0 / (double)v->y_dims);
  k.inv_A = (float)(1.0 / (double)v->antialias_samples);
  k.inv_max_depth = v->shader_id == RRT_SHADER_DEPTH ? (float)(1.0 / (double)v->max_depth) : 0.f;
  a.S = s->S; a.B = s->B; a.nwords = g.nwords;
  a.P = g.P; a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y; a.n_tiles = g.n_tiles;
  char* base = (char*)workspace;
  a.ps = (const float*)(base + g.ps_off);
  a.hdr = (const float*)(base + g.hdr_off);
  a.acc = (float*)(base + g.acc_off);
  a.acc_repl = g.acc_repl;
  a.split_log2 = 0;
  a.mask = (const unsigned long long*)(base + g.mask_off);
  a.heavy = (const unsigned long long*)(base + g.heavy_off);
  a.empty = (const unsigned long long*)(base + g.empty_off);
  a.lds_scene = g.lds_mode;
  a.bins_per_scene = (int)(g.n_bins / (s->B > 0 ? s->B : 1));
  a.blk_counts = (const unsigned*)(base + g.counts_off);
  a.n_bins = (int)g.n_bins;
  a.wgq = (unsigned long long*)(base + g.wgq_off);
  a.wgq_cnt = (unsigned*)(base + g.wgq_cnt_off);
  a.use_saved = 0;
  a.chunk = g.chunk;
  // weighted shares only where a favoured workgroup's share surely fits its zero queue (average share <= Q_CAP / 2)
  const bool weigh = (long long)g.n_bins * SEG * 2 <= (long long)Q_CAP * g.blocks_fwd && !(g.blocks_fwd & 1u);
  a.ztail = tune().ztail;
  a.zw_even = weigh ? tune().zw_even : 1;
  a.zw_odd = weigh ? tune().zw_odd : 1;
  a.n_waves = 0;
  a.jitter = v->jitter_mode == RRT_JITTER_EXPLICIT ? v->jitter : nullptr;
  a.image = nullptr;
  a.grad = nullptr;
  return a;
}

// k_prep: everything the render kernels read from the workspace.
int launch_prepare(const rrt_scene* s, const rrt_view* v, const KArgs& a, hipStream_t st) {
  if (a.n_bins <= 0) return RRT_OK;
  PrepArgs pa;
  pa.w2o = s->w2o; pa.w2o_bs = (long long)s->w2o_bstride;
  pa.mat = s->material; pa.mat_bs = (long long)s->material_bstride;
  pa.light = s->light; pa.kind = s->kind;
  pa.cam = v->cam; pa.cam_bs = (long long)v->cam_bstride;
  pa.ps = const_cast<float*>(a.ps); pa.hdr = const_cast<float*>(a.hdr); pa.acc = a.acc;
  pa.mask = const_cast<unsigned long long*>(a.mask);
  pa.heavy = const_cast<unsigned long long*>(a.heavy);
  pa.empty = const_cast<unsigned long long*>(a.empty);
  pa.blk_counts = const_cast<unsigned*>(a.blk_counts);
  hipLaunchKernelGGL(k_prep, dim3((unsigned)a.n_bins), dim3(256), 0, st, a, pa);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e);
  return RRT_OK;
}

template <int P, int JC, int TR>
void launch_fwd_jt(const KArgs& a, unsigned blocks, int shader, int spec, hipStream_t st) {
  // (extension) hard shadows: their own instantiations, generic jitter code only -- the default kernels are untouched
  const bool sh = (a.v.flags & RRT_FLAG_SHADOWS) != 0 && shader == SHADER_PHONG;
  if (shader == SHADER_DEPTH) hipLaunchKernelGGL((k_fwd<P, SHADER_DEPTH, 0, 0, TR, 0>), dim3(blocks), dim3(BLOCK), 0, st, a);
  else if (sh && spec) hipLaunchKernelGGL((k_fwd<P, SHADER_PHONG, 1, 0, TR, 1>), dim3(blocks), dim3(BLOCK), 0, st, a);
  else if (sh) hipLaunchKernelGGL((k_fwd<P, SHADER_PHONG, 0, 0, TR, 1>), dim3(blocks), dim3(BLOCK), 0, st, a);
  else if (spec) hipLaunchKernelGGL((k_fwd<P, SHADER_PHONG, 1, JC, TR, 0>), dim3(blocks), dim3(BLOCK), 0, st, a);
  else hipLaunchKernelGGL((k_fwd<P, SHADER_PHONG, 0, JC, TR, 0>), dim3(blocks), dim3(BLOCK), 0, st, a);
}
// JC = 1: jitter at the stratum centre (the per-ray hash / explicit-jitter code drops out of the row
// loops); TR = view.transpose (which grid coordinate runs along a wave's lanes)
template <int P>
void launch_fwd(const KArgs& a, unsigned blocks, int shader, int spec, hipStream_t st) {
  const bool jc = a.v.jitter_mode == JITTER_CENTER;
  if (a.v.transpose) { if (jc) launch_fwd_jt<P, 1, 1>(a, blocks, shader, spec, st); else launch_fwd_jt<P, 0, 1>(a, blocks, shader, spec, st); }
  else { if (jc) launch_fwd_jt<P, 1, 0>(a, blocks, shader, spec, st); else launch_fwd_jt<P, 0, 0>(a, blocks, shader, spec, st); }
}

template <int P, int JC, int TR>
void launch_bwd_jt(const KArgs& a, unsigned blocks, int shader, int spec, hipStream_t st) {
  const bool sh = (a.v.flags & RRT_FLAG_SHADOWS) != 0 && shader == SHADER_PHONG;
  if (shader == SHADER_DEPTH) hipLaunchKernelGGL((k_bwd<P, SHADER_DEPTH, 0, 0, TR, 0>), dim3(blocks), dim3(BLOCK), 0, st, a);
  else if (sh && spec) hipLaunchKernelGGL((k_bwd<P, SHADER_PHONG, 1, 0, TR, 1>), dim3(blocks), dim3(BLOCK), 0, st, a);
  else if (sh) hipLaunchKernelGGL((k_bwd<P, SHADER_PHONG, 0, 0, TR, 1>), dim3(blocks), dim3(BLOCK), 0, st, a);
  else if (P <= 2 && a.split_log2) {
    // small images with several samples per pixel: the samples of a tile are shared out between waves
    if (spec) hipLaunchKernelGGL((k_bwd<P <= 2 ? P : 1, SHADER_PHONG, 1, JC, TR, 0, 1>), dim3(blocks), dim3(BLOCK), 0, st, a);
    else hipLaunchKernelGGL((k_bwd<P <= 2 ? P : 1, SHADER_PHONG, 0, JC, TR, 0, 1>), dim3(blocks), dim3(BLOCK), 0, st, a);
  }
  else if (spec) hipLaunchKernelGGL((k_bwd<P, SHADER_PHONG, 1, JC, TR, 0>), dim3(blocks), dim3(BLOCK), 0, st, a);
  else hipLaunchKernelGGL((k_bwd<P, SHADER_PHONG, 0, JC, TR, 0>), dim3(blocks), dim3(BLOCK), 0, st, a);
}
template <int P>
void launch_bwd(const KArgs& a, unsigned blocks, int shader, int spec, hipStream_t st) {
  const bool jc = a.v.jitter_mode == JITTER_CENTER;
  if (a.v.transpose) { if (jc) launch_bwd_jt<P, 1, 1>(a, blocks, shader, spec, st); else launch_bwd_jt<P, 0, 1>(a, blocks, shader, spec, st); }
  else { if (jc) launch_bwd_jt<P, 1, 0>(a, blocks, shader, spec, st); else launch_bwd_jt<P, 0, 0>(a, blocks, shader, spec, st); }
}

}  // namespace

extern "C" {

#ifdef RRT_STAMPS
int rrt_debug_set_stamps(void* dev_ptr) {
  unsigned long long* p = (unsigned long long*)dev_ptr;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p));
}
#endif

// Tests only (not part of rrt.h): read the RRT_TUNE_* environment again.  Not safe against concurrent renders.
int rrt_debug_reload_tuning(void) {
  tune();
  read_tune(g_tune);
  return 0;
}

int rrt_version(void) { return RRT_VERSION; }

const char* rrt_strerror(int status) {
  switch (status) {
    case RRT_OK: return "ok";
    case RRT_ERR_NULL: return "required pointer is NULL";
    case RRT_ERR_SHAPE: return "invalid or inconsistent dimension";
    case RRT_ERR_TOO_MANY: return "too many shapes (S > RRT_MAX_SHAPES)";
    case RRT_ERR_WORKSPACE: return "workspace NULL, too small or not 16-byte aligned";
    case RRT_ERR_ENUM: return "unknown shader / jitter mode / shape kind";
    case RRT_ERR_HIP: return "HIP runtime error (see rrt_last_hip_error)";
    default: return "unknown rrt status";
  }
}

int rrt_last_hip_error(void) { return g_last_hip_error; }

int rrt_image_cols(const rrt_view* v) {
  if (!v) return RRT_ERR_NULL;
  return v->transpose ? v->y_dims : v->x_dims;
}

int rrt_image_rows_full(const rrt_view* v) {
  if (!v) return RRT_ERR_NULL;
  return v->transpose ? v->x_dims : v->y_dims;
}

size_t rrt_workspace_bytes(const rrt_scene* s, const rrt_view* v) {
  if (validate(s, v) != RRT_OK) return 0;
  return geometry_of(s, v).total;
}

int rrt_render_fwd(const rrt_scene* s, const rrt_view* v, float* image_out, void* workspace,
                   size_t workspace_bytes, void* hip_stream) {
  int rc = validate(s, v);
  if (rc != RRT_OK) return rc;
  Geometry g = geometry_of(s, v);
  if (!workspace || workspace_bytes < g.total || ((uintptr_t)workspace & 15)) return RRT_ERR_WORKSPACE;
  if (v->rows > 0 && !image_out) return RRT_ERR_NULL;
  hipStream_t st = (hipStream_t)hip_stream;
  KArgs a = make_args(s, v, workspace, g);
  a.image = image_out;
  if (!(v->flags & RRT_FLAG_REUSE_SETUP)) {
    rc = launch_prepare(s, v, a, st);
    if (rc != RRT_OK) return rc;
  }
  if (g.n_tiles == 0) return RRT_OK;
  unsigned blocks = g.blocks_fwd;
  a.n_waves = (int)blocks * WAVES;
  if (g.P == 4) launch_fwd<4>(a, blocks, a.v.shader, a.v.specular, st);
  else if (g.P == 2) launch_fwd<2>(a, blocks, a.v.shader, a.v.specular, st);
  else launch_fwd<1>(a, blocks, a.v.shader, a.v.specular, st);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RRT_OK : hip_fail(e);
}

int rrt_render_bwd(const rrt_scene* s, const rrt_view* v, const float* grad_image, float* g_w2o,
                   float* g_material, float* g_light, void* workspace, size_t workspace_bytes,
                   void* hip_stream) {
  int rc = validate(s, v);
  if (rc != RRT_OK) return rc;
  Geometry g = geometry_of(s, v);
  if (!workspace || workspace_bytes < g.total || ((uintptr_t)workspace & 15)) return RRT_ERR_WORKSPACE;
  if (v->rows > 0 && !grad_image) return RRT_ERR_NULL;
  hipStream_t st = (hipStream_t)hip_stream;
  KArgs a = make_args(s, v, workspace, g);
  a.grad = grad_image;
  if (!(v->flags & RRT_FLAG_REUSE_SETUP)) {
    rc = launch_prepare(s, v, a, st);
    if (rc != RRT_OK) return rc;
  }
  hipError_t e;
  if (g.n_tiles == 0) {
    // an empty band (a rank with no rows): nothing was prepared, the gradients are zero
    size_t n_shapes = (size_t)s->B * (size_t)s->S;
    e = hipSuccess;
    if (g_w2o && n_shapes) e = hipMemsetAsync(g_w2o, 0, n_shapes * 12 * sizeof(float), st);
    if (e == hipSuccess && g_material && n_shapes) e = hipMemsetAsync(g_material, 0, n_shapes * 7 * sizeof(float), st);
    if (e == hipSuccess && g_light) e = hipMemsetAsync(g_light, 0, 6 * sizeof(float), st);
    return e == hipSuccess ? RRT_OK : hip_fail(e);
  }
  if (s->S > 0) {
    unsigned blocks = g.blocks_bwd;
    a.n_waves = (int)blocks * WAVES;
    // the kernel that ran on this setup before (forward or backward) left its queue batches behind
    a.use_saved = ((v->flags & RRT_FLAG_REUSE_SETUP) && g.blocks_fwd == g.blocks_bwd && g.lds_mode != 2 && !tune().no_saved) ? 1 : 0;
    // Tiles of 1 or 2 rows mean fewer than 4 tiles per wave (geometry_of): with several samples per pixel a wave's serial
    // walk through the samples of its tiles is then the kernel's critical path, and the samples are shared out
    {
      int parts = tune().split >= 0 ? tune().split : (a.v.A >= 4 ? 4 : 2);
      if (parts > a.v.A) parts = a.v.A >= 2 ? 2 : 0;
      a.split_log2 = (g.P <= 2 && parts >= 2) ? (parts == 4 ? 2 : 1) : 0;
    }
    if (g.P == 4) launch_bwd<4>(a, blocks, a.v.shader, a.v.specular, st);
    else if (g.P == 2) launch_bwd<2>(a, blocks, a.v.shader, a.v.specular, st);
    else launch_bwd<1>(a, blocks, a.v.shader, a.v.specular, st);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e);
  }
  if (v->flags & RRT_FLAG_SKIP_FINALIZE) return RRT_OK;
#define RRT_LAUNCH_FINALIZE(R)                                                                                       \
  hipLaunchKernelGGL(k_finalize<R>, dim3(1), dim3(256), 0, st, s->S, s->B, a.acc, a.hdr, s->material,                  \
                     (long long)s->material_bstride, s->light, (long long)s->w2o_bstride, g_w2o, g_material, g_light)
  if (g.acc_repl == 8) RRT_LAUNCH_FINALIZE(8);
  else if (g.acc_repl == 4) RRT_LAUNCH_FINALIZE(4);
  else if (g.acc_repl == 2) RRT_LAUNCH_FINALIZE(2);
  else
    hipLaunchKernelGGL(k_finalize_one, dim3(1), dim3(256), 0, st, s->S, s->B, a.acc, a.hdr, s->material,
                       (long long)s->material_bstride, s->light, (long long)s->w2o_bstride, g_w2o, g_material, g_light);
#undef RRT_LAUNCH_FINALIZE
  e = hipGetLastError();
  return e == hipSuccess ? RRT_OK : hip_fail(e);
}

}  // extern "C"
