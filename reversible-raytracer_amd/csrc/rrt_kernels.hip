// rrt_kernels.hip -- gfx950 (MI355X, CDNA4) render kernels + the C ABI of rrt.h.
//
// Four kernels per forward/backward pair (DESIGN.md has the full story):
//
//   k_setup     one thread per (scene, shape): fold the camera into the shape
//               transform, precompute per-shape constants and a camera-local
//               bounding ball; one 128-B record per shape in the workspace.
//   k_fwd<P>    one workgroup (4 waves) per 64 x 4P pixel tile, one lane per
//               pixel column, each wave walks P rows.  Tile prologue: 64-wide
//               ballot cull of the shapes against the tile's ray cone -> a
//               candidate bit mask in LDS.  Per ray: rays are generated in
//               registers, candidate shapes are visited in list order with
//               their parameters read through wave-uniform SCALAR loads
//               (SGPR operands: no VGPRs, no LDS traffic), the winner is shaded
//               in a waterfall over the distinct winners of the wave.  The RGB
//               row is transposed through a wave-private LDS slab so the image
//               is written as three fully coalesced 256-B dword stores.
//   k_bwd<P>    same tiling / cull; tiles without candidates exit before
//               touching the upstream gradient.  Phase 1 finds each ray's
//               winner, phase 2 walks the winners wave-uniformly: lanes that
//               hit shape k run the analytic adjoint with k's parameters in
//               SGPRs and add into 22 per-lane accumulators, which are summed
//               across the wave with DPP (row_shr / row_bcast) and leave as ONE
//               22-lane global_atomic_add_f32 per (wave, sample, shape).
//   k_finalize  chain the accumulator rows back to w2o / material / light.
//
// No MFMA: the path is per-ray quadratic solves, not a contraction.
#include <hip/hip_runtime.h>

#include "rrt.h"
#include "rrt_device.h"

namespace {

using namespace rrt;

constexpr int TILE_W = 64;      // one lane per pixel column: a wave row is 768 contiguous bytes
constexpr int WAVES = 4;
constexpr int BLOCK = WAVES * 64;
constexpr int MAX_WORDS = RRT_MAX_SHAPES / 64;

// Wave-uniform read-only data goes through the constant address space so the
// backend selects s_load_* (scalar cache, SGPR results).
typedef const __attribute__((address_space(4))) float* cfptr;
__device__ __forceinline__ cfptr as_const(const float* p) { return (cfptr)p; }

struct KArgs {
  ViewK v;
  int S, B, nwords;
  int tiles_x, tiles_y;
  const float* ps;      // [B][S][PS_STRIDE]
  const float* hdr;     // [B][HDR_STRIDE]
  const float* jitter;  // explicit sampleDist or nullptr
  float* image;         // fwd: [B][rows][ncols][3]
  const float* grad;    // bwd: same layout
  float* acc;           // bwd: [B][S][ACC_STRIDE]
};

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long x) {
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// ------------------------------------------------------------------ setup --
__global__ void k_setup(int S, int B, const float* w2o, long long w2o_bs, const float* mat, long long mat_bs,
                        const float* light, const uint8_t* kind, const float* cam, long long cam_bs,
                        float* ps, float* hdr) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < B * S) {
    int b = idx / S, k = idx - b * S;
    pack_shape<float, float>(w2o + b * w2o_bs + 12 * k, cam + b * cam_bs, mat + b * mat_bs + 7 * k, light,
                             kind ? (int)kind[k] : 0, ps + (size_t)idx * PS_STRIDE);
  }
  if (idx < B) pack_header<float, float>(cam + idx * cam_bs, light, hdr + (size_t)idx * HDR_STRIDE);
}

// ------------------------------------------------------------ tile helpers --
struct Tile {
  int b, a0, b0, a1, b1;   // scene, first/last row (window-relative) and column
};

template <int P>
__device__ __forceinline__ Tile decode_tile(const KArgs& a) {
  Tile t;
  int tile = blockIdx.x;
  int tx = tile % a.tiles_x;
  int rest = tile / a.tiles_x;
  int ty = rest % a.tiles_y;
  t.b = rest / a.tiles_y;
  t.a0 = ty * (WAVES * P);
  t.b0 = tx * TILE_W;
  t.a1 = min(t.a0 + WAVES * P, a.v.rows) - 1;
  t.b1 = min(t.b0 + TILE_W, a.v.ncols) - 1;
  return t;
}

// Candidate shapes of the tile: 64 shapes per ballot, one word per wave pass.
// Returns the OR of all words (wave-uniform).  Ends with a workgroup barrier.
__device__ __forceinline__ unsigned long long cull_tile(const KArgs& a, const Tile& t,
                                                        unsigned long long* s_mask) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int i0, j0, i1, j1;
  pixel_to_ray(a.v, t.a0 + a.v.row0, t.b0, i0, j0);
  pixel_to_ray(a.v, t.a1 + a.v.row0, t.b1, i1, j1);
  Cone<float> cone = tile_cone<float>(a.v, i0, i1, j0, j1);
  for (int w = wave; w < a.nwords; w += WAVES) {
    int k = w * 64 + lane;
    bool c = false;
    if (k < a.S) {
      if (a.v.flags & RRT_FLAG_NO_CULL) {
        c = true;
      } else {
        const float* p = a.ps + ((size_t)t.b * a.S + k) * PS_STRIDE;
        float4 cr = *reinterpret_cast<const float4*>(p + PS_CULL_C);
        float cc[3] = {cr.x, cr.y, cr.z};
        c = cone_may_hit<float>(cone, cc, cr.w);
      }
    }
    unsigned long long m = __ballot(c);
    if (lane == 0) s_mask[w] = m;
  }
  __syncthreads();
  unsigned long long any = 0;
  for (int w = 0; w < a.nwords; ++w) any |= s_mask[w];
  return uniform_u64(any);
}

// Winner of the composite for one ray (scene.py:38-47): candidates in list
// order, strict '<', first wins ties.  Shape parameters are wave-uniform.
__device__ __forceinline__ int trace_ray(const KArgs& a, cfptr ps, const unsigned long long* s_mask,
                                         const float d[3], float& tbest, float nu[3]) {
  int best = -1;
  float tmin = INFINITY;
  for (int w = 0; w < a.nwords; ++w) {
    unsigned long long m = uniform_u64(s_mask[w]);
    while (m) {
      int kb = __builtin_ctzll(m);
      m &= m - 1;
      int k = w * 64 + kb;
      cfptr p = ps + (size_t)k * PS_STRIDE;
      float A[9] = {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]};
      float o[3] = {p[PS_O], p[PS_O + 1], p[PS_O + 2]};
      int kind = ps_load_int(p[PS_KIND]);
      float r[3];
      xform_dir<float>(A, d, r);
      if (kind == KIND_SPHERE) {
        SphereHit<float> sp;
        bool hit = hit_sphere<float>(r, o, sp);
        if (hit && sp.t < tmin) {
          tmin = sp.t; best = k;
          sphere_point<float>(sp, r, nu);
        }
      } else {
        float t;
        bool hit = hit_square<float>(r, o, t);
        if (hit && t < tmin) {
          tmin = t; best = k;
          nu[0] = 0.f; nu[1] = 0.f; nu[2] = o[2] > 0.f ? 1.f : -1.f;
        }
      }
    }
  }
  tbest = tmin;
  return best;
}

// ---------------------------------------------------------------- forward --
template <int P>
__global__ __launch_bounds__(BLOCK) void k_fwd(KArgs a) {
  __shared__ unsigned long long s_mask[MAX_WORDS];
  __shared__ float s_stage[WAVES][3 * TILE_W];
  const ViewK& v = a.v;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const Tile t = decode_tile<P>(a);
  const unsigned long long any = cull_tile(a, t, s_mask);

  cfptr ps = as_const(a.ps) + (size_t)t.b * a.S * PS_STRIDE;
  cfptr hdr = as_const(a.hdr) + (size_t)t.b * HDR_STRIDE;
  const float L[3] = {hdr[HDR_LHAT], hdr[HDR_LHAT + 1], hdr[HDR_LHAT + 2]};
  const float look[3] = {hdr[HDR_LOOK], hdr[HDR_LOOK + 1], hdr[HDR_LOOK + 2]};

  const int col = t.b0 + lane;
  const bool col_ok = col <= t.b1;
  const int nvalid = 3 * (t.b1 - t.b0 + 1);
  float* st = s_stage[wave];

#pragma unroll 1
  for (int p = 0; p < P; ++p) {
    const int row = t.a0 + wave + WAVES * p;       // wave-uniform
    if (row > t.a1) break;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (any) {
      int i, j;
      pixel_to_ray(v, row + v.row0, col_ok ? col : t.b1, i, j);
#pragma unroll 1
      for (int s = 0; s < v.A; ++s) {
        float sdx, sdy, d[3];
        sample_dist<float>(v, a.jitter, i, j, s, sdx, sdy);
        ray_dir<float>(v, i, j, sdx, sdy, d);
        float tb, nu[3] = {0.f, 0.f, 1.f};
        int best = trace_ray(a, ps, s_mask, d, tb, nu);
        if (v.shader == SHADER_DEPTH) {
          if (best >= 0) {
            float c = shade_depth<float>(tb, v.inv_max_depth);
            c0 += c; c1 += c; c2 += c;
          }
        } else {
          // waterfall over the distinct winners of this wave
          bool todo = best >= 0;
          while (true) {
            unsigned long long vote = __ballot(todo);
            if (!vote) break;
            int kk = __builtin_amdgcn_readlane(best, __builtin_ctzll(vote));
            if (todo && best == kk) {
              cfptr q = ps + (size_t)kk * PS_STRIDE;
              float colI[3] = {q[PS_COLI], q[PS_COLI + 1], q[PS_COLI + 2]};
              float c[3];
              shade_phong<float>(nu, colI, q[PS_KS], q[PS_KD], q[PS_KA], q[PS_SH], ps_load_int(q[PS_POWF]),
                                 v.specular, L, look, c);
              c0 += c[0]; c1 += c[1]; c2 += c[2];
              todo = false;
            }
          }
        }
      }
      c0 *= v.inv_A; c1 *= v.inv_A; c2 *= v.inv_A;
    }
    // RGB row -> wave-private LDS slab -> three coalesced dword stores
    st[3 * lane + 0] = c0;
    st[3 * lane + 1] = c1;
    st[3 * lane + 2] = c2;
    __builtin_amdgcn_wave_barrier();
    float* out = a.image + (((size_t)t.b * v.rows + row) * v.ncols + t.b0) * 3;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      int idx = lane + 64 * q;
      float val = st[idx];
      if (idx < nvalid) out[idx] = val;
    }
    __builtin_amdgcn_wave_barrier();
  }
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

template <int P>
__global__ __launch_bounds__(BLOCK) void k_bwd(KArgs a) {
  __shared__ unsigned long long s_mask[MAX_WORDS];
  __shared__ float s_stage[WAVES][3 * TILE_W];
  const ViewK& v = a.v;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const Tile t = decode_tile<P>(a);
  const unsigned long long any = cull_tile(a, t, s_mask);
  if (!any) return;                                // no shape can be hit: nothing to read, nothing to add

  cfptr ps = as_const(a.ps) + (size_t)t.b * a.S * PS_STRIDE;
  cfptr hdr = as_const(a.hdr) + (size_t)t.b * HDR_STRIDE;
  const float L[3] = {hdr[HDR_LHAT], hdr[HDR_LHAT + 1], hdr[HDR_LHAT + 2]};
  const float look[3] = {hdr[HDR_LOOK], hdr[HDR_LOOK + 1], hdr[HDR_LOOK + 2]};

  const int col = t.b0 + lane;
  const bool col_ok = col <= t.b1;
  const int nvalid = 3 * (t.b1 - t.b0 + 1);
  float* st = s_stage[wave];

  // upstream gradient of this lane's P pixels (coalesced dword loads -> LDS -> per-pixel RGB)
  float up[P][3];
  bool live[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int row = t.a0 + wave + WAVES * p;
    const bool row_ok = row <= t.a1;               // wave-uniform
    live[p] = row_ok && col_ok;
    up[p][0] = up[p][1] = up[p][2] = 0.f;
    if (row_ok) {
      const float* in = a.grad + (((size_t)t.b * v.rows + row) * v.ncols + t.b0) * 3;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        int idx = lane + 64 * q;
        st[idx] = idx < nvalid ? in[idx] : 0.f;
      }
      __builtin_amdgcn_wave_barrier();
      up[p][0] = st[3 * lane + 0] * v.inv_A;
      up[p][1] = st[3 * lane + 1] * v.inv_A;
      up[p][2] = st[3 * lane + 2] * v.inv_A;
      __builtin_amdgcn_wave_barrier();
    }
  }

  float* acc_scene = a.acc + (size_t)t.b * a.S * ACC_STRIDE;

#pragma unroll 1
  for (int s = 0; s < v.A; ++s) {
    // phase 1: every ray's winner
    float d[P][3];
    int best[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int row = t.a0 + wave + WAVES * p;
      int i, j;
      pixel_to_ray(v, min(row, t.a1) + v.row0, col_ok ? col : t.b1, i, j);
      float sdx, sdy;
      sample_dist<float>(v, a.jitter, i, j, s, sdx, sdy);
      ray_dir<float>(v, i, j, sdx, sdy, d[p]);
      float tb, nu[3];
      int k = trace_ray(a, ps, s_mask, d[p], tb, nu);
      best[p] = live[p] ? k : -1;
    }
    // phase 2: wave-uniform walk over the candidate shapes that won somewhere
    for (int w = 0; w < a.nwords; ++w) {
      unsigned long long m = uniform_u64(s_mask[w]);
      while (m) {
        int kb = __builtin_ctzll(m);
        m &= m - 1;
        int k = w * 64 + kb;
        bool mine = false;
#pragma unroll
        for (int p = 0; p < P; ++p) mine |= (best[p] == k);
        if (!__ballot(mine)) continue;
        cfptr q = ps + (size_t)k * PS_STRIDE;
        float A[9] = {q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8]};
        float o[3] = {q[PS_O], q[PS_O + 1], q[PS_O + 2]};
        float colI[3] = {q[PS_COLI], q[PS_COLI + 1], q[PS_COLI + 2]};
        float ks = q[PS_KS], kd = q[PS_KD], ka = q[PS_KA], sh = q[PS_SH];
        int kind = ps_load_int(q[PS_KIND]), powf = ps_load_int(q[PS_POWF]);
        float acc[ACC_USED];
#pragma unroll
        for (int n = 0; n < ACC_USED; ++n) acc[n] = 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if (best[p] == k)
            adjoint_ray<float>(kind, A, o, colI, ks, kd, ka, sh, powf, v.shader, v.specular, v.inv_max_depth,
                               L, look, d[p], up[p], acc);
        }
        // 22 wave sums -> lane n holds value n -> one 22-lane atomic
        float mine_val = 0.f;
#pragma unroll
        for (int n = 0; n < ACC_USED; ++n) {
          float tot = wave_sum_to_lane63(acc[n]);
          float bc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), 63));
          mine_val = (lane == n) ? bc : mine_val;
        }
        if (lane < ACC_USED)
          __hip_atomic_fetch_add(acc_scene + (size_t)k * ACC_STRIDE + lane, mine_val, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// --------------------------------------------------------------- finalize --
// grid: one block of 256 threads; loops are tiny (B*S rows of 32 floats).
__global__ __launch_bounds__(256) void k_finalize(int S, int B, const float* acc, const float* hdr,
                                                  const float* mat, long long mat_bs, const float* light,
                                                  long long w2o_bs, float* g_w2o, float* g_mat, float* g_light) {
  __shared__ float s_red[256][6];
  const int tid = threadIdx.x;
  const int Bw = w2o_bs ? B : 1, Bm = mat_bs ? B : 1;
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
    for (int n = 0; n < 6; ++n) s_red[tid][n] = gl[n];
    __syncthreads();
    for (int stride = 128; stride > 0; stride >>= 1) {
      if (tid < stride)
        for (int n = 0; n < 6; ++n) s_red[tid][n] += s_red[tid + stride][n];
      __syncthreads();
    }
    if (tid == 0) {
      float g[6];
      for (int n = 0; n < 6; ++n) g[n] = s_red[0][n];
      finalize_light_dir<float>(hdr, g);            // the light is shared by all scenes
      for (int n = 0; n < 6; ++n) g_light[n] = g[n];
    }
  }
}

// ------------------------------------------------------------------- host --
thread_local int g_last_hip_error = 0;

int hip_fail(hipError_t e) {
  g_last_hip_error = (int)e;
  return RRT_ERR_HIP;
}

struct Layout {
  size_t ps_off, hdr_off, acc_off, total;
};

Layout layout_of(int S, int B) {
  Layout l;
  l.ps_off = 0;
  size_t ps_bytes = (size_t)B * (size_t)S * PS_STRIDE * sizeof(float);
  l.hdr_off = ps_bytes;
  l.acc_off = l.hdr_off + (size_t)B * HDR_STRIDE * sizeof(float);
  l.total = l.acc_off + (size_t)B * (size_t)S * ACC_STRIDE * sizeof(float);
  if (l.total == 0) l.total = 16;
  return l;
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
  return RRT_OK;
}

// P = 4 (64x16 tiles) once the launch has enough tiles to fill the chip,
// otherwise P = 1 (64x4 tiles) to expose more workgroups.
int choose_P(const rrt_scene* s, int rows, int ncols) {
  long long tx = (ncols + TILE_W - 1) / TILE_W;
  long long ty4 = (rows + 4 * WAVES - 1) / (4 * WAVES);
  return (tx * ty4 * s->B >= 2048) ? 4 : 1;
}

KArgs make_args(const rrt_scene* s, const rrt_view* v, void* workspace, int P) {
  KArgs a;
  ViewK& k = a.v;
  k.x_dims = v->x_dims; k.y_dims = v->y_dims; k.A = v->antialias_samples;
  k.row0 = v->row0; k.rows = v->rows;
  k.ncols = v->transpose ? v->y_dims : v->x_dims;
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
  a.S = s->S; a.B = s->B; a.nwords = (s->S + 63) / 64;
  a.tiles_x = (k.ncols + TILE_W - 1) / TILE_W;
  a.tiles_y = (k.rows + WAVES * P - 1) / (WAVES * P);
  Layout l = layout_of(s->S, s->B);
  char* base = (char*)workspace;
  a.ps = (const float*)(base + l.ps_off);
  a.hdr = (const float*)(base + l.hdr_off);
  a.acc = (float*)(base + l.acc_off);
  a.jitter = v->jitter_mode == RRT_JITTER_EXPLICIT ? v->jitter : nullptr;
  a.image = nullptr;
  a.grad = nullptr;
  return a;
}

int launch_setup(const rrt_scene* s, const rrt_view* v, const KArgs& a, hipStream_t st) {
  int n = s->B * s->S > s->B ? s->B * s->S : s->B;
  int blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_setup, dim3(blocks), dim3(256), 0, st, s->S, s->B, s->w2o, (long long)s->w2o_bstride,
                     s->material, (long long)s->material_bstride, s->light, s->kind, v->cam,
                     (long long)v->cam_bstride, const_cast<float*>(a.ps), const_cast<float*>(a.hdr));
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RRT_OK : hip_fail(e);
}

}  // namespace

extern "C" {

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
  return layout_of(s->S, s->B).total;
}

int rrt_render_fwd(const rrt_scene* s, const rrt_view* v, float* image_out, void* workspace,
                   size_t workspace_bytes, void* hip_stream) {
  int rc = validate(s, v);
  if (rc != RRT_OK) return rc;
  if (v->rows == 0) return RRT_OK;
  if (!image_out) return RRT_ERR_NULL;
  Layout l = layout_of(s->S, s->B);
  if (!workspace || workspace_bytes < l.total || ((uintptr_t)workspace & 15)) return RRT_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)hip_stream;
  int ncols = v->transpose ? v->y_dims : v->x_dims;
  int P = choose_P(s, v->rows, ncols);
  KArgs a = make_args(s, v, workspace, P);
  a.image = image_out;
  if (!(v->flags & RRT_FLAG_REUSE_SETUP)) {
    rc = launch_setup(s, v, a, st);
    if (rc != RRT_OK) return rc;
  }
  long long tiles = (long long)a.tiles_x * a.tiles_y * s->B;
  if (tiles > 0x7fffffffLL) return RRT_ERR_SHAPE;
  if (P == 4) hipLaunchKernelGGL(k_fwd<4>, dim3((unsigned)tiles), dim3(BLOCK), 0, st, a);
  else hipLaunchKernelGGL(k_fwd<1>, dim3((unsigned)tiles), dim3(BLOCK), 0, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RRT_OK : hip_fail(e);
}

int rrt_render_bwd(const rrt_scene* s, const rrt_view* v, const float* grad_image, float* g_w2o,
                   float* g_material, float* g_light, void* workspace, size_t workspace_bytes,
                   void* hip_stream) {
  int rc = validate(s, v);
  if (rc != RRT_OK) return rc;
  Layout l = layout_of(s->S, s->B);
  if (!workspace || workspace_bytes < l.total || ((uintptr_t)workspace & 15)) return RRT_ERR_WORKSPACE;
  if (v->rows > 0 && !grad_image) return RRT_ERR_NULL;
  hipStream_t st = (hipStream_t)hip_stream;
  int ncols = v->transpose ? v->y_dims : v->x_dims;
  int P = choose_P(s, v->rows, ncols);
  KArgs a = make_args(s, v, workspace, P);
  a.grad = grad_image;
  if (!(v->flags & RRT_FLAG_REUSE_SETUP)) {
    rc = launch_setup(s, v, a, st);
    if (rc != RRT_OK) return rc;
  }
  hipError_t e;
  if (s->S > 0) {
    e = hipMemsetAsync(a.acc, 0, (size_t)s->B * s->S * ACC_STRIDE * sizeof(float), st);
    if (e != hipSuccess) return hip_fail(e);
  }
  long long tiles = (long long)a.tiles_x * a.tiles_y * s->B;
  if (tiles > 0x7fffffffLL) return RRT_ERR_SHAPE;
  if (tiles > 0 && s->S > 0) {
    if (P == 4) hipLaunchKernelGGL(k_bwd<4>, dim3((unsigned)tiles), dim3(BLOCK), 0, st, a);
    else hipLaunchKernelGGL(k_bwd<1>, dim3((unsigned)tiles), dim3(BLOCK), 0, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e);
  }
  hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, st, s->S, s->B, a.acc, a.hdr, s->material,
                     (long long)s->material_bstride, s->light, (long long)s->w2o_bstride, g_w2o, g_material,
                     g_light);
  e = hipGetLastError();
  return e == hipSuccess ? RRT_OK : hip_fail(e);
}

}  // extern "C"
