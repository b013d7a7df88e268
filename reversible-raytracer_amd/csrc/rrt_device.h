// rrt_device.h -- per-ray / per-shape math of the render hot path.
//
// Everything here is a small inline function over scalars so that the HIP
// kernels (rrt_kernels.hip) keep wave-uniform shape parameters in SGPRs and
// per-ray state in VGPRs.  The functions are templated on the real type and
// compile on the host as well (plain g++), which is how the CPU unit tests
// (tests/cpu_harness) check the hand-derived adjoint in float64 against the
// autograd oracle without a GPU.  The shipped library only instantiates float.
//
// Reference semantics implemented (file:line in lebek/reversible-raytracer):
//   ray generation          scene.py:61-75      (rrt::ray_dir)
//   ray transform           transform.py:40-47  (folded into the packed A', o')
//   sphere intersection     shape.py:78-83,109-126 (rrt::hit_sphere)
//   sphere normals          shape.py:128-138    (object-space, kept as is)
//   square intersection     shape.py:25-69      (rrt::hit_square)
//   Phong shading           shader.py:28-53     (rrt::shade_phong)
//   depth shading           shader.py:14-20     (rrt::shade_depth)
//   adjoint of all of it    T.grad, optimize.py:25 (rrt::adjoint_ray) -- not
//                           written anywhere in the reference; derived in
//                           DESIGN.md section "Adjoint"
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RRT_HD __host__ __device__ __forceinline__
#else
#define RRT_HD inline
#endif

namespace rrt {

// ---------------------------------------------------------------- layouts --
// Packed shape record, written by the setup kernel, read (wave-uniformly)
// by the render kernels.  32 floats = one 128-byte line per shape.
constexpr int PS_STRIDE = 32;
constexpr int PS_A = 0;       // [9]  A' = w2o_linear * cam_linear, row-major
constexpr int PS_O = 9;       // [3]  o' = w2o * cam_origin
constexpr int PS_GAMMA = 12;  //      o'.o' - 1
constexpr int PS_COLI = 13;   // [3]  color * light intensity
constexpr int PS_KS = 16;
constexpr int PS_KD = 17;
constexpr int PS_KA = 18;
constexpr int PS_SH = 19;
constexpr int PS_KIND = 20;   //      RRT_KIND_* (ps_store_int / ps_load_int)
constexpr int PS_POWF = 21;   //      POW_* / FAST_* flags (ps_store_int / ps_load_int)
constexpr int PS_CMAX = 22;   //      max_j colI_j           (fast paths: one compare finds a wave without upper clipping)
constexpr int PS_KSSH = 23;   //      ks * shininess         (fast adjoint)
constexpr int PS_CULL_C = 24; // [3]  centre of a bounding ball, camera-local
constexpr int PS_CULL_R = 27; //      its radius (inf = never cull)
constexpr int PS_ELL = 28;    // [3]  w2o_linear . (-Lhat): the direction towards the light in object space (shadows)
constexpr int PS_ELL2 = 31;   //      its squared length

// Per-scene header (one per batch element).
constexpr int HDR_STRIDE = 32;
constexpr int HDR_LHAT = 0;   // [3]  normed light direction (scene.py:83-86)
constexpr int HDR_LOOK = 3;   // [3]  camera.look_at
constexpr int HDR_INVLN = 6;  //      1/|L|
constexpr int HDR_LLOOK = 7;  //      Lhat . look_at   (s = rm . look_at = Lhat.look_at + 2 dd (n . look_at))
constexpr int HDR_CAM = 8;    // [9]  camera o2w linear part, row-major
constexpr int HDR_CAMO = 17;  // [3]  camera origin

// Gradient accumulator row per (scene, shape).
constexpr int ACC_STRIDE = 32;
constexpr int ACC_RD = 0;     // [9]  sum g_r' (x) d_local
constexpr int ACC_O = 9;      // [3]  sum g_o'
constexpr int ACC_COLI = 12;  // [3]  sum masked-upstream * phi
constexpr int ACC_KA = 15;
constexpr int ACC_KD = 16;
constexpr int ACC_KS = 17;
constexpr int ACC_SH = 18;
constexpr int ACC_LHAT = 19;  // [3]
constexpr int ACC_USED = 22;

constexpr int KIND_SPHERE = 0;
constexpr int KIND_SQUARE = 1;
constexpr int SHADER_PHONG = 0;
constexpr int SHADER_DEPTH = 1;
constexpr int JITTER_CENTER = 0;
constexpr int JITTER_EXPLICIT = 1;
constexpr int JITTER_HASH = 2;

constexpr int POW_INT = 1;   // shininess is integer valued
constexpr int POW_ODD = 2;   // ... and odd
// A shape whose render can take the lean code paths (shade_phong_fast / adjoint_sphere_phong):
// a Sphere with colour * intensity > 0 on every channel (then clip(phi colI_j, 0, 1) is clipped
// at 0 on ALL channels exactly when phi < 0); with the specular term also shininess > 1 (then
// |s|^(sh-1) is finite at s = 0 and the special cases of pow() at exponent 0 and 1 do not arise).
constexpr int FAST_NOSPEC = 4;
constexpr int FAST_SPEC = 8;

// ------------------------------------------------------------ math shims --
template <class T> RRT_HD T m_sqrt(T x) { return sqrt(x); }
template <class T> RRT_HD T m_rcp(T x) { return T(1) / x; }
template <class T> RRT_HD T m_rsqrt(T x) { return T(1) / sqrt(x); }
template <class T> RRT_HD T m_log2(T x) { return log2(x); }
template <class T> RRT_HD T m_exp2(T x) { return exp2(x); }
template <class T> RRT_HD T m_abs(T x) { return x < T(0) ? -x : x; }
// a*b + c with ONE rounding, spelled out wherever the forward image depends on it: left to the compiler,
// which product of "x*y + z*w" gets fused depends on where the operands live (SGPR or VGPR), and the
// same pixel would differ in its last bit between the single- and the multi-candidate code path
RRT_HD float m_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RRT_HD double m_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <class T> RRT_HD T dot3(const T a[3], const T b[3]) { return m_fma(a[2], b[2], m_fma(a[1], b[1], a[0] * b[0])); }
template <class T> RRT_HD T m_inf() { return T(INFINITY); }
template <class T> RRT_HD T m_nan() { return T(NAN); }

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 hardware transcendentals (v_sqrt_f32, v_rcp_f32, v_rsq_f32,
// v_log_f32, v_exp_f32: 1 ulp, quarter rate).
template <> RRT_HD float m_sqrt<float>(float x) { return __builtin_amdgcn_sqrtf(x); }
template <> RRT_HD float m_rcp<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <> RRT_HD float m_rsqrt<float>(float x) { return __builtin_amdgcn_rsqf(x); }
template <> RRT_HD float m_log2<float>(float x) { return __builtin_amdgcn_logf(x); }
template <> RRT_HD float m_exp2<float>(float x) { return __builtin_amdgcn_exp2f(x); }
template <> RRT_HD float m_abs<float>(float x) { return __builtin_fabsf(x); }
#endif

// |x| if c else x ; e (>= 0 or NaN) with the sign of s if c else e.  c is wave-uniform on the GPU, where
// both are one bit operation with a scalar mask instead of a compare + select.
template <class T> RRT_HD T abs_if(T x, bool c) { return c ? m_abs<T>(x) : x; }
template <class T> RRT_HD T sign_of_if(T e, T s, bool c) { return (c && s < T(0)) ? -e : e; }
#if defined(__HIP_DEVICE_COMPILE__)
template <> RRT_HD float abs_if<float>(float x, bool c) {
  return __uint_as_float(__float_as_uint(x) & (c ? 0x7fffffffu : 0xffffffffu));
}
template <> RRT_HD float sign_of_if<float>(float e, float s, bool c) {
  return __uint_as_float(__float_as_uint(e) | (__float_as_uint(s) & (c ? 0x80000000u : 0u)));
}
// does the predicate hold on any active lane of the wave?
#define RRT_WAVE_ANY(c) (__ballot(c) != 0ull)
#else
#define RRT_WAVE_ANY(c) (c)
#endif

// Small ints inside the packed float record: bit pattern for float storage (so
// a wave-uniform scalar load yields the int without a VALU conversion), plain
// number for the float64 storage of the CPU test harness.
RRT_HD float ps_store_int(int i, float) { union { int i; float f; } c; c.i = i; return c.f; }
RRT_HD double ps_store_int(int i, double) { return (double)i; }
RRT_HD int ps_load_int(float f) { union { int i; float f; } c; c.f = f; return c.i; }
RRT_HD int ps_load_int(double f) { return (int)f; }

// --------------------------------------------------------------- the view --
// Launch-uniform view constants (kernel argument, lives in SGPRs).
template <class T>
struct ViewT {
  int x_dims, y_dims, A;
  int row0, rows, ncols;    // output window and its width
  int transpose;            // 1: image[a,b] <- ray[i=b, j=a]; 0: ray[i=a, j=b]
  int jitter_mode;
  uint32_t seed;
  int shader, specular;
  uint32_t flags;
  T ustep, vstep;           // linspace steps: u(i) = .5 + i*ustep, v(j) = -.5 + j*vstep
  T inv_xd, inv_yd;         // 1/x_dims, 1/y_dims
  T inv_A;
  T inv_max_depth;
};
using ViewK = ViewT<float>;   // what the kernels use (float64 only in the CPU test harness)

// Counter hash -> U in [0,1).  Mirrors oracle/rrt_oracle.py:hash_u01 bit for bit.
RRT_HD float hash_u01(uint32_t seed, int x_dims, int A, int i, int j, int s, int axis) {
  uint32_t idx = (((uint32_t)i * (uint32_t)x_dims + (uint32_t)j) * (uint32_t)A + (uint32_t)s) * 2u + (uint32_t)axis;
  uint32_t h = idx ^ (seed * 0x9E3779B9u);
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// sampleDist_x, sampleDist_y of scene.py:31-32 for ray (i,j), sample s.
template <class T>
RRT_HD void sample_dist(const ViewT<T>& v, const float* jitter, int i, int j, int s, T& sdx, T& sdy) {
  if (v.jitter_mode == JITTER_EXPLICIT) {
    const float* p = jitter + ((((size_t)i * (size_t)v.x_dims + (size_t)j) * (size_t)v.A + (size_t)s) * 2);
    sdx = T(p[0]);
    sdy = T(p[1]);
  } else if (v.jitter_mode == JITTER_HASH) {
    sdx = T((hash_u01(v.seed, v.x_dims, v.A, i, j, s, 0) + (float)s) * (float)v.inv_A);
    sdy = T((hash_u01(v.seed, v.x_dims, v.A, i, j, s, 1) + (float)s) * (float)v.inv_A);
  } else {
    sdx = sdy = T((0.5f + (float)s) * (float)v.inv_A);
  }
}

// Camera.make_rays (scene.py:66-74): d = normalize(u_i, v_j, 1) then
// d.x += sdx/x_dims, d.y += sdy/y_dims (jitter AFTER normalisation).
//
// Split so that the kernels keep what depends only on a lane's own grid coordinate out of
// their row loops: with cl the coordinate that varies along image columns (u under the root
// camera's transpose, w otherwise) and cr the one that varies along rows,
//   1/|(u, w, 1)| = rsqrt(cr^2 + (cl^2 + 1)),   q = cl^2 + 1 is per lane and tile.
// jx, jy = sampleDist / dims.  Every caller (kernels, CPU harness) goes through ray_dir_parts,
// so a ray has the same bits whichever kernel, tile shape or row band generates it.
template <class T>
RRT_HD T grid_u(const ViewT<T>& v, int i) { return m_fma(T(i), T(v.ustep), T(0.5)); }
template <class T>
RRT_HD T grid_w(const ViewT<T>& v, int j) { return m_fma(T(j), T(v.vstep), T(-0.5)); }

template <class T>
RRT_HD void ray_dir_parts(T u, T w, T q, T cr, T jx, T jy, T d[3]) {
  T inv = m_rsqrt<T>(m_fma(cr, cr, q));
  d[0] = m_fma(u, inv, jx);
  d[1] = m_fma(w, inv, jy);
  d[2] = inv;
}

template <class T>
RRT_HD void ray_dir(const ViewT<T>& v, int i, int j, T sdx, T sdy, T d[3]) {
  T u = grid_u<T>(v, i), w = grid_w<T>(v, j);
  T cl = v.transpose ? u : w, cr = v.transpose ? w : u;
  ray_dir_parts<T>(u, w, m_fma(cl, cl, T(1)), cr, sdx * T(v.inv_xd), sdy * T(v.inv_yd), d);
}

// Output pixel (a,b) of the image window -> ray grid index (i,j).
template <class T>
RRT_HD void pixel_to_ray(const ViewT<T>& v, int a, int b, int& i, int& j) {
  if (v.transpose) { i = b; j = a; } else { i = a; j = b; }
}

// ---------------------------------------------------------- intersection --
// Object-space ray direction r' = A' d.
template <class T>
RRT_HD void xform_dir(const T* A, const T d[3], T r[3]) {
  r[0] = dot3<T>(A, d);
  r[1] = dot3<T>(A + 3, d);
  r[2] = dot3<T>(A + 6, d);
}

// Sphere.distance (shape.py:109-126): t = (-beta - sqrt(Delta))/alpha if
// Delta = beta^2 - alpha*gamma > 0 else miss.  No t > 0 test.
//
// Evaluated in the algebraically identical, better conditioned form
//   q = beta/alpha,  o_perp = o' - q r'  (foot of the perpendicular from the
//   sphere centre to the ray, object space),  D' = 1 - |o_perp|^2 = Delta/alpha,
//   h = sqrt(D'/alpha) = sqrt(Delta)/alpha,  t = -q - h
// because float32 beta^2 - alpha*gamma cancels |o'|^2-fold (|o'| ~ 10..100 for
// the small far spheres of the bench scene) right where the gradient divides
// by sqrt(Delta).  The float64 oracle evaluates the reference's own formula.
template <class T>
struct SphereHit {
  T alpha, beta, ia, q, op[3], Dp, h, t;
  T ih;       // 1/h (hit_sphere<T, true> only)
};

// WITH_IH (the adjoint): h and 1/h from ONE reciprocal square root (h = x rsqrt(x)) instead of a
// square root here and a reciprocal later -- transcendental ops issue at a quarter of the FP32
// rate and each one less is ~1 us of k_bwd.  The forward kernels keep the plain square root.
template <class T, bool WITH_IH = false>
RRT_HD bool hit_sphere(const T r[3], const T o[3], SphereHit<T>& sh) {
  sh.alpha = dot3<T>(r, r);
  sh.beta = dot3<T>(r, o);
  sh.ia = m_rcp<T>(sh.alpha);
  sh.q = sh.beta * sh.ia;
  sh.op[0] = m_fma(-sh.q, r[0], o[0]);
  sh.op[1] = m_fma(-sh.q, r[1], o[1]);
  sh.op[2] = m_fma(-sh.q, r[2], o[2]);
  sh.Dp = T(1) - dot3<T>(sh.op, sh.op);
  bool hit = sh.Dp > T(0);            // false for NaN, as the reference's mask
  const T x = (hit ? sh.Dp : T(1)) * sh.ia;
  if (WITH_IH) {
    sh.ih = m_rsqrt<T>(x);
    sh.h = x * sh.ih;
  } else {
    sh.h = m_sqrt<T>(x);
    sh.ih = T(0);
  }
  sh.t = -sh.q - sh.h;
  return hit;
}

// Hit point in object space, p = o' + t r' = o_perp - h r' (shape.py:134).
template <class T>
RRT_HD void sphere_point(const SphereHit<T>& sh, const T r[3], T p[3]) {
  p[0] = m_fma(-sh.h, r[0], sh.op[0]);
  p[1] = m_fma(-sh.h, r[1], sh.op[1]);
  p[2] = m_fma(-sh.h, r[2], sh.op[2]);
}

// Square._hit (shape.py:25-40): t = -o_z/r_z, hit iff r_z != 0, t > 0 and the
// hit point is strictly inside (-.5,.5)^2.
template <class T>
RRT_HD bool hit_square(const T r[3], const T o[3], T& t) {
  bool np = r[2] != T(0);
  t = -o[2] * m_rcp<T>(np ? r[2] : T(1));
  T x = m_fma(t, r[0], o[0]);
  T y = m_fma(t, r[1], o[1]);
  return np && (t > T(0)) && (x > T(-0.5)) && (x < T(0.5)) && (y > T(-0.5)) && (y < T(0.5));
}

// ---------------------------------------------------------------- shadows --
// EXTENSION, default off (RRT_FLAG_SHADOWS), no reference behaviour to be identical to: the hard
// shadows of the reference's dormant Sphere.shadow (shape.py:85-97; its call site scene.py:41-45 is
// commented out).  The point a ray (camera-local direction d, distance t) hits, in the object space
// of ANOTHER Sphere with packed A2, o2: y = o2 + t A2 d.  It is in that sphere's shadow iff the ray
// from y towards the light, y + tau ell (ell = w2o_linear . (-Lhat), not unit in general), enters the
// unit sphere ahead of it: discriminant (y.ell)^2 - |ell|^2 (|y|^2 - 1) > 0 and the nearer root
// tau = (-(y.ell) - sqrt(disc)) / |ell|^2 >= 0  <=>  y.ell < 0 and |y| >= 1 (no square root needed).
// For ell = -Lhat (a pure translation) this is Sphere.shadow's "-x - sqrt(decider) >= 0".
template <class T>
RRT_HD bool shadowed_by(const T* A2, const T o2[3], const T ell[3], T ell2, const T d[3], T t) {
  T r2[3];
  xform_dir<T>(A2, d, r2);
  const T y[3] = {o2[0] + t * r2[0], o2[1] + t * r2[1], o2[2] + t * r2[2]};
  const T yl = y[0] * ell[0] + y[1] * ell[1] + y[2] * ell[2];
  const T yy = y[0] * y[0] + y[1] * y[1] + y[2] * y[2];
  const T disc = yl * yl - ell2 * (yy - T(1));
  return disc > T(0) && yl < T(0) && yy >= T(1);
}

// --------------------------------------------------------------- shading --
// x ** y with C pow() semantics for the cases a render can produce
// (shader.py:45): negative base is finite for integer-valued y, NaN otherwise.
// Returns |x|^y with sign; also |x|^(y-1) with sign through pw1 and log2|x| through l2
// (both for the adjoint).
template <class T>
RRT_HD T pow_signed(T x, T y, int flags, T& pw1, T& l2) {
  T a = m_abs<T>(x);
  l2 = m_log2<T>(a);
  T p = (y == T(0)) ? T(1) : m_exp2<T>(y * l2);
  T p1 = (y == T(1)) ? T(1) : m_exp2<T>((y - T(1)) * l2);
  if (x < T(0)) {
    if (flags & POW_INT) {
      if (flags & POW_ODD) p = -p; else p1 = -p1;
    } else {
      p = m_nan<T>();
      p1 = m_nan<T>();
    }
  }
  pw1 = p1;
  return p;
}

// The adjoint's variant: |x|^(y-1) by exp2 and |x|^y = |x|^(y-1) |x| (one exp2 instead of two).
template <class T>
RRT_HD T pow_signed_adj(T x, T y, int flags, T& pw1, T& l2) {
  T a = m_abs<T>(x);
  l2 = m_log2<T>(a);
  T p1 = (y == T(1)) ? T(1) : m_exp2<T>((y - T(1)) * l2);
  T p = (y == T(0)) ? T(1) : (a > T(0) ? p1 * a : T(0));   // (0^y = 0 also where 0^(y-1) = inf)
  if (x < T(0)) {
    if (flags & POW_INT) {
      if (flags & POW_ODD) p = -p; else p1 = -p1;
    } else {
      p = m_nan<T>();
      p1 = m_nan<T>();
    }
  }
  pw1 = p1;
  return p;
}

// Everything shading needs about the winning hit, kept per ray.
template <class T>
struct Hit {
  T t;        // distance along the ray
  T nu[3];    // un-normalised normal: sphere p = o' + t r' (shape.py:134); square (0,0,+-1) (shape.py:52-69)
};

// PhongShader.shade (shader.py:28-53) for one hit.  mat = colI[3], ks, kd, ka, sh.
template <class T>
RRT_HD void shade_phong(const T nu[3], const T colI[3], T ks, T kd, T ka, T sh, int powf,
                        int specular, const T L[3], const T look[3], T c[3]) {
  T inv = m_rsqrt<T>(dot3<T>(nu, nu));
  const T n[3] = {nu[0] * inv, nu[1] * inv, nu[2] * inv};
  T dd = -dot3<T>(n, L);                                // normals . (-Lhat), NOT clamped
  T phi = m_fma(kd, dd, ka);
  if (specular) {
    T two = T(2) * dd;
    const T rm[3] = {m_fma(two, n[0], L[0]), m_fma(two, n[1], L[1]), m_fma(two, n[2], L[2])};
    T s = dot3<T>(rm, look);
    T pw1, l2;
    phi = m_fma(ks, pow_signed<T>(s, sh, powf, pw1, l2), phi);
  }
  for (int q = 0; q < 3; ++q) {
    T x = phi * colI[q];
    c[q] = x < T(0) ? T(0) : (x > T(1) ? T(1) : x);       // NaN passes through like T.clip
  }
}

// The same for a FAST_* shape (colI > 0 on every channel; with the specular term shininess > 1),
// leaner: s = rm . look_at = Lhat.look_at + 2 dd (n . look_at) instead of building rm; pow() as one
// log2 / exp2 with the sign handled by bit masks (no exponent-0 / exponent-1 / zero-base cases to
// select between: with shininess > 0 a zero base gives exp2(-inf) = 0).
template <class T, int SPEC>
RRT_HD void shade_phong_fast(const T nu[3], const T colI[3], T ks, T kd, T ka, T sh, int powf,
                             const T L[3], const T look[3], T Llook, T c[3]) {
  T inv = m_rsqrt<T>(dot3<T>(nu, nu));
  const T n[3] = {nu[0] * inv, nu[1] * inv, nu[2] * inv};
  T ndl = dot3<T>(n, L);                                // dd = -ndl
  T phi = m_fma(-kd, ndl, ka);
  if (SPEC) {
    T nlk = dot3<T>(n, look);
    T s = m_fma(-(ndl + ndl), nlk, Llook);
    T e = m_exp2<T>(sh * m_log2<T>(abs_if<T>(s, (powf & POW_INT) != 0)));   // non-integer shininess: log2 of s < 0 is NaN, as pow()
    phi = m_fma(ks, sign_of_if<T>(e, s, (powf & POW_ODD) != 0), phi);
  }
  for (int q = 0; q < 3; ++q) {
    T x = phi * colI[q];
    c[q] = x < T(0) ? T(0) : (x > T(1) ? T(1) : x);       // NaN passes through like T.clip
  }
}

// DepthMapShader.shade (shader.py:14-20): 1 - t/maxDepth on all channels.
template <class T>
RRT_HD T shade_depth(T t, T inv_max_depth) { return m_fma(-t, inv_max_depth, T(1)); }

// --------------------------------------------------------------- adjoint --
// Reverse-mode of one ray that hit shape `kind` (the composite winner), given
// the upstream gradient up[3] = dL/d(colour of this sample) (already divided
// by antialias_samples).  Accumulates into acc[ACC_STRIDE] (see ACC_*).
//
// Inputs are the packed shape parameters (wave-uniform on the GPU) and the
// camera-local ray direction d.  Returns whether the ray hit (false: nothing
// was added).  Derivation: DESIGN.md "Adjoint".
template <class T>
RRT_HD bool adjoint_ray(int kind, const T* A, const T o[3],
                        const T colI[3], T ks, T kd, T ka, T sh, int powf,
                        int shader, int specular, T inv_max_depth,
                        const T L[3], const T look[3],
                        const T d[3], const T up[3], T* acc) {
  T r[3];
  xform_dir<T>(A, d, r);
  T t;
  SphereHit<T> sp;
  bool hit;
  if (kind == KIND_SPHERE) { hit = hit_sphere<T, true>(r, o, sp); t = sp.t; }
  else hit = hit_square<T>(r, o, t);
  if (!hit) return false;                                   // miss => exactly zero

  T g_t = T(0);
  T g_p[3] = {T(0), T(0), T(0)};
  if (shader == SHADER_DEPTH) {
    g_t = -(up[0] + up[1] + up[2]) * inv_max_depth;
  } else {
    T nu[3];
    if (kind == KIND_SPHERE) {
      sphere_point<T>(sp, r, nu);
    } else {
      nu[0] = T(0); nu[1] = T(0); nu[2] = (o[2] > T(0)) ? T(1) : T(-1);
    }
    T inv = m_rsqrt<T>(nu[0] * nu[0] + nu[1] * nu[1] + nu[2] * nu[2]);
    T n[3] = {nu[0] * inv, nu[1] * inv, nu[2] * inv};
    T dd = -(n[0] * L[0] + n[1] * L[1] + n[2] * L[2]);
    T phi = ka + kd * dd;
    T s = T(0), pw = T(0), pw1 = T(0), l2s = T(0);
    if (specular) {
      T two = T(2) * dd;
      s = (two * n[0] + L[0]) * look[0] + (two * n[1] + L[1]) * look[1] + (two * n[2] + L[2]) * look[2];
      pw = pow_signed_adj<T>(s, sh, powf, pw1, l2s);
      phi += ks * pw;
    }
    // clip(phi*colI, 0, 1): gradient passes where 0 <= x <= 1 (inclusive, as
    // Theano's Clip.grad and torch.clamp)
    T g_phi = T(0);
    for (int q = 0; q < 3; ++q) {
      T x = phi * colI[q];
      T um = (x >= T(0) && x <= T(1)) ? up[q] : T(0);
      acc[ACC_COLI + q] += um * phi;
      g_phi += um * colI[q];
    }
    acc[ACC_KA] += g_phi;
    acc[ACC_KD] += g_phi * dd;
    T g_dd = g_phi * kd;
    T g_n[3] = {T(0), T(0), T(0)};
    T g_L[3] = {T(0), T(0), T(0)};
    if (specular) {
      acc[ACC_KS] += g_phi * pw;
      // d/dshininess = pw * ln(s); only defined for s > 0 (extension, see rrt.h)
      if (s > T(0)) acc[ACC_SH] += g_phi * ks * pw * (l2s * T(0.6931471805599453));   // l2s = log2(s) from the pow
      T g_s = g_phi * ks * sh * pw1;
      T g_rm[3] = {g_s * look[0], g_s * look[1], g_s * look[2]};
      // rm = 2 dd n + Lhat
      g_dd += T(2) * (g_rm[0] * n[0] + g_rm[1] * n[1] + g_rm[2] * n[2]);
      T two = T(2) * dd;
      for (int q = 0; q < 3; ++q) { g_n[q] = two * g_rm[q]; g_L[q] = g_rm[q]; }
    }
    // dd = -(n . Lhat)
    for (int q = 0; q < 3; ++q) { g_n[q] -= g_dd * L[q]; g_L[q] -= g_dd * n[q]; }
    acc[ACC_LHAT + 0] += g_L[0];
    acc[ACC_LHAT + 1] += g_L[1];
    acc[ACC_LHAT + 2] += g_L[2];
    if (kind == KIND_SPHERE) {
      // n = p/|p|
      T ndg = n[0] * g_n[0] + n[1] * g_n[1] + n[2] * g_n[2];
      for (int q = 0; q < 3; ++q) g_p[q] = (g_n[q] - n[q] * ndg) * inv;
      g_t = g_p[0] * r[0] + g_p[1] * r[1] + g_p[2] * r[2];
    }
    // square: the normal is piecewise constant -> nothing reaches geometry
  }

  // p = o' + t r'  (sphere, Phong);  then t = t(r', o')
  T g_o[3] = {g_p[0], g_p[1], g_p[2]};
  T g_r[3] = {t * g_p[0], t * g_p[1], t * g_p[2]};
  if (kind == KIND_SPHERE) {
    // t = (-beta - sqrt(Delta))/alpha, Delta = beta^2 - alpha*gamma, gamma = o'.o' - 1:
    //   g_beta = -g_t/alpha + 2 beta g_D,  g_alpha = -g_t t/alpha - gamma g_D,
    //   g_gamma = -alpha g_D,              g_D = -g_t / (2 alpha sqrt(Delta))
    // collected so that only well conditioned quantities appear
    // (beta o' - gamma r' = beta o_perp + D' r';  o' - (beta/alpha) r' = o_perp):
    T gta = g_t * sp.ia;                              // g_t / alpha
    T g_D = T(-0.5) * gta * sp.ia * sp.ih;            // sqrt(Delta) = alpha h
    T two_gD = T(2) * g_D;
    T ca = two_gD * sp.alpha;
    for (int q = 0; q < 3; ++q) {
      g_o[q] += -gta * r[q] - ca * sp.op[q];
      g_r[q] += -gta * o[q] - T(2) * gta * t * r[q] + two_gD * (sp.beta * sp.op[q] + sp.Dp * r[q]);
    }
  } else {
    // t = -o_z / r_z
    T irz = m_rcp<T>(r[2]);
    g_o[2] += -g_t * irz;
    g_r[2] += -g_t * t * irz;
  }
  for (int q = 0; q < 3; ++q) {
    acc[ACC_RD + 3 * q + 0] += g_r[q] * d[0];
    acc[ACC_RD + 3 * q + 1] += g_r[q] * d[1];
    acc[ACC_RD + 3 * q + 2] += g_r[q] * d[2];
    acc[ACC_O + q] += g_o[q];
  }
  return true;
}

// The hot path of the adjoint: a FAST_* Sphere under the Phong shader (see FAST_NOSPEC / FAST_SPEC).
// Same mathematics as adjoint_ray, organised for the VALU:
//   * colI > 0  =>  every channel is clipped at 0 exactly when phi < 0: such a ray (and a NaN phi,
//     whose clip masks are all false) contributes nothing and leaves with the misses; the upper
//     clip is tested once per wave on the brightest channel (cmax) and only a wave that has a
//     saturated lane builds per-channel masks;
//   * s = Lhat.look + 2 dd (n.look): rm is never built, and g_rm . n = g_s (n.look) reuses n.look;
//   * |s|^(sh-1) by one log2 / exp2, sign by bit masks, |s|^sh = |s|^(sh-1) s  (shininess > 1: no
//     special cases; non-integer shininess with s < 0 gives NaN like pow());
//   * g_n = c1 look - g_dd Lhat and G_Lhat = g_s look - g_dd n straight from scalars.
// kssh = ks * shininess, shm1 = shininess - 1, Llook = Lhat . look_at, cmax = max_j colI_j.
// *gs_sum collects sum g_s: the caller adds gs_sum * look_at to acc[ACC_LHAT..] when its rows are done
// (adjoint_sphere_phong_finish).  Returns whether the ray contributed.
template <class T, int SPEC>
RRT_HD bool adjoint_sphere_phong(const T* A, const T o[3], const T colI[3], T cmax, T ks, T kd, T ka, T shm1, T kssh,
                                 int powf, const T L[3], const T look[3], T Llook, const T d[3], const T up[3],
                                 T* acc, T* gs_sum) {
  T r[3];
  xform_dir<T>(A, d, r);
  const T alpha = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  const T beta = r[0] * o[0] + r[1] * o[1] + r[2] * o[2];
  const T ia = m_rcp<T>(alpha);
  const T q = beta * ia;
  const T op[3] = {o[0] - q * r[0], o[1] - q * r[1], o[2] - q * r[2]};
  const T Dp = m_fma(-op[2], op[2], m_fma(-op[1], op[1], m_fma(-op[0], op[0], T(1))));
  // a miss (Dp <= 0) needs no test of its own: 1/sqrt of a non-positive number is NaN or infinite, h = x / sqrt(x)
  // and with it phi become NaN, and the ONE test below drops the ray -- one exec-mask region per row instead of two
  const T x = Dp * ia;
  const T ih = m_rsqrt<T>(x);
  const T h = x * ih;
  const T t = -q - h;
  // The hit point p = o' + t r' = o_perp - h r' lies ON the unit sphere -- t is the root of |o' + t r'| = 1 -- so the
  // reference's normals p / |p| (shape.py:137) divide by exactly 1: the adjoint uses n = p and |p| = 1 (float32 puts |p|
  // within ~1e-5 of 1; the forward image keeps the division).  Saves a reciprocal square root and nine multiplies a row.
  const T n[3] = {op[0] - h * r[0], op[1] - h * r[1], op[2] - h * r[2]};
  const T ndl = n[0] * L[0] + n[1] * L[1] + n[2] * L[2];      // dd = -ndl
  T phi = ka - kd * ndl;
  T nlk = T(0), s = T(0), p1 = T(0), pw = T(0), l2 = T(0);
  if (SPEC) {
    nlk = n[0] * look[0] + n[1] * look[1] + n[2] * look[2];
    s = Llook - (ndl + ndl) * nlk;
    l2 = m_log2<T>(abs_if<T>(s, (powf & POW_INT) != 0));
    p1 = sign_of_if<T>(m_exp2<T>(shm1 * l2), s, (powf & (POW_INT | POW_ODD)) == POW_INT);   // sign(s)^(sh-1)
    pw = p1 * s;
    phi += ks * pw;
  }
  // a miss, all three channels clipped at 0, or NaN: exactly zero
  if (!(phi >= T(0)) || !(Dp > T(0))) return false;
  T um[3] = {up[0], up[1], up[2]};
  if (RRT_WAVE_ANY(phi * cmax > T(1))) {
    for (int c = 0; c < 3; ++c) um[c] = (phi * colI[c] <= T(1)) ? up[c] : T(0);
  }
  const T g_phi = um[0] * colI[0] + um[1] * colI[1] + um[2] * colI[2];
  acc[ACC_COLI + 0] += um[0] * phi;
  acc[ACC_COLI + 1] += um[1] * phi;
  acc[ACC_COLI + 2] += um[2] * phi;
  acc[ACC_KA] += g_phi;
  acc[ACC_KD] -= g_phi * ndl;
  T g_dd = g_phi * kd;
  T g_n[3], ndg;
  if (SPEC) {
    acc[ACC_KS] += g_phi * pw;
    // d/dshininess = pw ln(s), lanes with s <= 0 contribute 0 (extension, see rrt.h)
    const T l2p = (s > T(0)) ? l2 : T(0);
    acc[ACC_SH] += (g_phi * ks) * pw * (l2p * T(0.6931471805599453));
    const T g_s = (g_phi * kssh) * p1;
    const T g_s2 = g_s + g_s;
    g_dd += g_s2 * nlk;                                       // rm = 2 dd n + Lhat:  g_dd += 2 g_rm . n,  g_rm = g_s look
    const T c1 = -ndl * g_s2;                                 // g_n = 2 dd g_rm - g_dd Lhat
    for (int c = 0; c < 3; ++c) {
      g_n[c] = c1 * look[c] - g_dd * L[c];
      acc[ACC_LHAT + c] -= g_dd * n[c];                       // G_Lhat += g_rm - g_dd n; the g_rm = g_s look part is summed as a
    }                                                         // scalar (gs_sum) and applied once per tile by the caller
    *gs_sum += g_s;
    ndg = c1 * nlk - g_dd * ndl;                              // n . g_n from the scalars at hand
  } else {
    for (int c = 0; c < 3; ++c) {
      g_n[c] = -g_dd * L[c];
      acc[ACC_LHAT + c] -= g_dd * n[c];
    }
    ndg = -g_dd * ndl;
  }
  // n = p/|p|, |p| = 1
  const T g_p[3] = {g_n[0] - n[0] * ndg, g_n[1] - n[1] * ndg, g_n[2] - n[2] * ndg};      // (I - n n^T) g_n
  const T g_t = g_p[0] * r[0] + g_p[1] * r[1] + g_p[2] * r[2];
  // t = t(r', o'), collected as in adjoint_ray (only well conditioned quantities appear)
  const T gta = g_t * ia;
  const T gD2 = -gta * ia * ih;                               // 2 g_D,  sqrt(Delta) = alpha h
  const T ca = gD2 * alpha;
  // g_r' = t g_p - gta o' + (gD2 D' - 2 gta t) r' + gD2 beta o_perp, with o' = o_perp + q r' (how o_perp was made):
  const T cr = gD2 * Dp - gta * (t + t + q);
  const T cop = gD2 * beta - gta;
  for (int c = 0; c < 3; ++c) {
    const T g_o = g_p[c] - gta * r[c] - ca * op[c];
    const T g_r = t * g_p[c] + cr * r[c] + cop * op[c];
    acc[ACC_RD + 3 * c + 0] += g_r * d[0];
    acc[ACC_RD + 3 * c + 1] += g_r * d[1];
    acc[ACC_RD + 3 * c + 2] += g_r * d[2];
    acc[ACC_O + c] += g_o;
  }
  return true;
}

template <class T>
RRT_HD void adjoint_sphere_phong_finish(T gs_sum, const T look[3], T* acc) {
  for (int c = 0; c < 3; ++c) acc[ACC_LHAT + c] += gs_sum * look[c];
}

// Which of the two a (shape, view) pair takes; the kernels and the CPU harness both dispatch on it.
RRT_HD bool adjoint_is_fast(int kind, int powf, int shader, int specular) {
  return kind == KIND_SPHERE && shader == SHADER_PHONG && (powf & (specular ? FAST_SPEC : FAST_NOSPEC)) != 0;
}

// ------------------------------------------------------------- tile cull --
// A double cone (lines through the camera origin) that contains every ray
// direction of a rectangle of the ray grid, camera-local.
template <class T>
struct Cone {
  T ax, ay, az;   // unit axis
  T cos_t, sin_t; // half angle; cos_t <= 0 means "everything"
};

// i in [i0,i1], j in [j0,j1] inclusive.  Jitter may move a unit direction by
// at most sqrt(inv_xd^2 + inv_yd^2) (sampleDist in [0,1]).
template <class T>
RRT_HD Cone<T> tile_cone(const ViewT<T>& v, int i0, int i1, int j0, int j1) {
  T u0 = T(0.5) + T(i0) * T(v.ustep), u1 = T(0.5) + T(i1) * T(v.ustep);
  T w0 = T(-0.5) + T(j0) * T(v.vstep), w1 = T(-0.5) + T(j1) * T(v.vstep);
  T uc = T(0.5) * (u0 + u1), wc = T(0.5) * (w0 + w1);
  T inv = m_rsqrt<T>(uc * uc + wc * wc + T(1));
  Cone<T> c;
  c.ax = uc * inv; c.ay = wc * inv; c.az = inv;
  T cmin = T(1);
  const T us[2] = {u0, u1};
  const T ws[2] = {w0, w1};
  for (int p = 0; p < 2; ++p)
    for (int q = 0; q < 2; ++q) {
      T ic = m_rsqrt<T>(us[p] * us[p] + ws[q] * ws[q] + T(1));
      T cs = (c.ax * us[p] + c.ay * ws[q] + c.az) * ic;
      cmin = cs < cmin ? cs : cmin;
    }
  cmin -= T(1e-6);
  if (cmin < T(0)) cmin = T(0);
  T smin = m_sqrt<T>(T(1) - cmin * cmin);
  T sj = m_sqrt<T>(T(v.inv_xd) * T(v.inv_xd) + T(v.inv_yd) * T(v.inv_yd)) * T(1.01) + T(1e-6);
  if (v.jitter_mode == JITTER_EXPLICIT) sj *= T(1);   // caller guarantees sampleDist in [0,1]
  if (sj >= T(1)) { c.cos_t = T(-1); c.sin_t = T(0); return c; }
  T cj = m_sqrt<T>(T(1) - sj * sj);
  c.cos_t = cmin * cj - smin * sj;
  c.sin_t = smin * cj + cmin * sj;
  return c;
}

// May any line of the cone touch the ball (centre cc, radius R)?  Conservative:
// false only when provably no ray of the tile can hit anything inside the ball.
template <class T>
RRT_HD bool cone_may_hit(const Cone<T>& c, const T cc[3], T R) {
  if (!(c.cos_t > T(0))) return true;
  T d2 = cc[0] * cc[0] + cc[1] * cc[1] + cc[2] * cc[2];
  T R1 = R * T(1.001);
  if (!(d2 > R1 * R1)) return true;                 // origin inside the ball, or NaN/inf
  T id = m_rsqrt<T>(d2);
  T sp = R1 * id;                                   // sin(psi) < 1
  T cp = m_sqrt<T>(T(1) - sp * sp);
  T cos_sum = c.cos_t * cp - c.sin_t * sp;          // cos(theta + psi)
  if (!(cos_sum > T(0))) return true;
  T cphi = m_abs<T>(cc[0] * c.ax + cc[1] * c.ay + cc[2] * c.az) * id;
  return !(cphi < cos_sum - T(1e-5));
}

// The same tile as a four-sided double pyramid: its rays are lines through the camera origin
// with directions (u, w, 1), u in [u0,u1], w in [w0,w1].  A wave tile is 64 x P pixels, so the
// circular cone above is up to 16x wider than the tile across its short side; the side planes
// are what keeps shapes a few rows away out of the candidate set.
template <class T>
struct Frustum {
  T u0, u1, w0, w1;       // widened by the reach of the jitter
  T nu0, nu1, nw0, nw1;   // 1/|normal| of the planes x = u z and y = w z
};

template <class T>
RRT_HD Frustum<T> tile_frustum(const ViewT<T>& v, int i0, int i1, int j0, int j1) {
  T ua = T(0.5) + T(i0) * T(v.ustep), ub = T(0.5) + T(i1) * T(v.ustep);
  T wa = T(-0.5) + T(j0) * T(v.vstep), wb = T(-0.5) + T(j1) * T(v.vstep);
  Frustum<T> f;
  f.u0 = ua < ub ? ua : ub; f.u1 = ua < ub ? ub : ua;
  f.w0 = wa < wb ? wa : wb; f.w1 = wa < wb ? wb : wa;
  // d = (u,w,1)/L + (sx/x_dims, sy/y_dims, 0), sx,sy in [0,1]  is the line (u + sx L/x_dims, w + sy L/y_dims, 1)
  T um = m_abs<T>(f.u0) > m_abs<T>(f.u1) ? f.u0 : f.u1, wm = m_abs<T>(f.w0) > m_abs<T>(f.w1) ? f.w0 : f.w1;
  T L = m_sqrt<T>(T(1) + um * um + wm * wm);
  T eu = T(v.inv_xd) * L * T(1.01) + T(1e-6), ew = T(v.inv_yd) * L * T(1.01) + T(1e-6);
  f.u0 -= eu; f.u1 += eu; f.w0 -= ew; f.w1 += ew;
  f.nu0 = m_rsqrt<T>(T(1) + f.u0 * f.u0); f.nu1 = m_rsqrt<T>(T(1) + f.u1 * f.u1);
  f.nw0 = m_rsqrt<T>(T(1) + f.w0 * f.w0); f.nw1 = m_rsqrt<T>(T(1) + f.w1 * f.w1);
  return f;
}

// May any line of the pyramid touch the ball?  Conservative: false only when the ball AND its
// mirror image (lines have no t > 0 test) each lie wholly outside one face of the half pyramid
// z >= 0.  NaN / infinite radii never cull.
template <class T>
RRT_HD bool frustum_may_hit(const Frustum<T>& f, const T cc[3], T R) {
  const T x = cc[0], y = cc[1], z = cc[2];
  const T R1 = R * T(1.001) + T(1e-5) * (m_abs<T>(x) + m_abs<T>(y) + m_abs<T>(z) + T(1));
  const T a1 = (x - f.u1 * z) * f.nu1, a0 = (f.u0 * z - x) * f.nu0;     // > R1: outside that face
  const T b1 = (y - f.w1 * z) * f.nw1, b0 = (f.w0 * z - y) * f.nw0;
  const bool out_pos = a1 > R1 || a0 > R1 || b1 > R1 || b0 > R1 || -z > R1;
  const bool out_neg = -a1 > R1 || -a0 > R1 || -b1 > R1 || -b0 > R1 || z > R1;
  return !(out_pos && out_neg);
}

// ---------------------------------------------------- setup and finalize --
// Pack one shape for one scene: fold the camera into the shape transform,
// precompute the per-shape constants and a bounding ball for the tile cull.
//   w2o[12]  rows 0..2 of shape.w2o.m           (shape.py:75)
//   cam[16]  rows 0..2 of camera o2w.m, look_at (orbit scene.py:56-61)
//   mat[7]   r,g,b,ks,kd,ka,shininess           (scene.py:90-101)
template <class T, class ST>
RRT_HD void pack_shape(const float* w2o, const float* cam, const float* mat, const float* light,
                       int kind, ST* ps) {
  T A[9], b[3], C[9], oc[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) { A[3 * r + c] = T(w2o[4 * r + c]); C[3 * r + c] = T(cam[4 * r + c]); }
    b[r] = T(w2o[4 * r + 3]);
    oc[r] = T(cam[4 * r + 3]);
  }
  T Ap[9], op[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c)
      Ap[3 * r + c] = A[3 * r] * C[c] + A[3 * r + 1] * C[3 + c] + A[3 * r + 2] * C[6 + c];
    op[r] = A[3 * r] * oc[0] + A[3 * r + 1] * oc[1] + A[3 * r + 2] * oc[2] + b[r];
  }
  for (int q = 0; q < 9; ++q) ps[PS_A + q] = (ST)Ap[q];
  for (int q = 0; q < 3; ++q) ps[PS_O + q] = (ST)op[q];
  ps[PS_GAMMA] = (ST)(op[0] * op[0] + op[1] * op[1] + op[2] * op[2] - T(1));
  for (int q = 0; q < 3; ++q) ps[PS_COLI + q] = (ST)(T(mat[q]) * T(light[3 + q]));
  ps[PS_KS] = (ST)mat[3]; ps[PS_KD] = (ST)mat[4]; ps[PS_KA] = (ST)mat[5]; ps[PS_SH] = (ST)mat[6];
  int powf = 0;
  float sh = mat[6];
  if (sh == floorf(sh) && fabsf(sh) < 16777216.0f) {
    powf |= POW_INT;
    if (((int)fabsf(sh)) & 1) powf |= POW_ODD;
  } else if (fabsf(sh) >= 16777216.0f && sh == sh) {
    powf |= POW_INT;   // every float this large is an even integer
  }
  {
    const T c0 = T(mat[0]) * T(light[3]), c1 = T(mat[1]) * T(light[4]), c2 = T(mat[2]) * T(light[5]);
    const T cmax = c0 > c1 ? (c0 > c2 ? c0 : c2) : (c1 > c2 ? c1 : c2);
    if (kind == KIND_SPHERE && c0 > T(0) && c1 > T(0) && c2 > T(0) && cmax < T(1e30)) {
      powf |= FAST_NOSPEC;
      if (sh > 1.0f && sh < 1e30f) powf |= FAST_SPEC;
    }
    ps[PS_CMAX] = (ST)cmax;
    ps[PS_KSSH] = (ST)(T(mat[3]) * T(mat[6]));
  }
  ps[PS_KIND] = ps_store_int(kind, ST(0));
  ps[PS_POWF] = ps_store_int(powf, ST(0));
  // bounding ball in camera-local space: x = A'^-1 (y - o'), |y| <= rho
  T c00 = Ap[4] * Ap[8] - Ap[5] * Ap[7], c01 = Ap[5] * Ap[6] - Ap[3] * Ap[8], c02 = Ap[3] * Ap[7] - Ap[4] * Ap[6];
  T det = Ap[0] * c00 + Ap[1] * c01 + Ap[2] * c02;
  T idet = T(1) / det;
  T I[9];
  I[0] = c00 * idet; I[1] = (Ap[2] * Ap[7] - Ap[1] * Ap[8]) * idet; I[2] = (Ap[1] * Ap[5] - Ap[2] * Ap[4]) * idet;
  I[3] = c01 * idet; I[4] = (Ap[0] * Ap[8] - Ap[2] * Ap[6]) * idet; I[5] = (Ap[2] * Ap[3] - Ap[0] * Ap[5]) * idet;
  I[6] = c02 * idet; I[7] = (Ap[1] * Ap[6] - Ap[0] * Ap[7]) * idet; I[8] = (Ap[0] * Ap[4] - Ap[1] * Ap[3]) * idet;
  T n1 = T(0), ninf = T(0);   // |I|_1 (max column sum), |I|_inf (max row sum); |I|_2 <= sqrt(|I|_1 |I|_inf)
  for (int q = 0; q < 3; ++q) {
    T cs = m_abs<T>(I[q]) + m_abs<T>(I[3 + q]) + m_abs<T>(I[6 + q]);
    T rs = m_abs<T>(I[3 * q]) + m_abs<T>(I[3 * q + 1]) + m_abs<T>(I[3 * q + 2]);
    n1 = cs > n1 ? cs : n1;
    ninf = rs > ninf ? rs : ninf;
  }
  T rho = (kind == KIND_SQUARE) ? T(0.70710678118654757) : T(1);
  T R = sqrt(n1 * ninf) * rho * T(1.001);
  T cc[3];
  for (int r = 0; r < 3; ++r) cc[r] = -(I[3 * r] * op[0] + I[3 * r + 1] * op[1] + I[3 * r + 2] * op[2]);
  bool ok = (R == R) && (R < T(1e30)) && (cc[0] == cc[0]) && (cc[1] == cc[1]) && (cc[2] == cc[2]);
  // float32 inversion of a badly conditioned A' is not trustworthy: never cull then
  T scaleA = T(0);
  for (int q = 0; q < 9; ++q) scaleA = m_abs<T>(Ap[q]) > scaleA ? m_abs<T>(Ap[q]) : scaleA;
  if (!(m_abs<T>(det) > T(1e-4) * scaleA * scaleA * scaleA)) ok = false;
  ps[PS_CULL_C + 0] = ok ? (ST)cc[0] : ST(0);
  ps[PS_CULL_C + 1] = ok ? (ST)cc[1] : ST(0);
  ps[PS_CULL_C + 2] = ok ? (ST)cc[2] : ST(0);
  ps[PS_CULL_R] = ok ? (ST)R : (ST)INFINITY;
  {
    // towards the light, in object space (shadows): w2o_linear . (-L / |L|)
    const T L0 = T(light[0]), L1 = T(light[1]), L2 = T(light[2]);
    const T il = T(-1) / sqrt(L0 * L0 + L1 * L1 + L2 * L2);
    T e[3];
    for (int r = 0; r < 3; ++r) e[r] = (A[3 * r] * L0 + A[3 * r + 1] * L1 + A[3 * r + 2] * L2) * il;
    ps[PS_ELL] = (ST)e[0]; ps[PS_ELL + 1] = (ST)e[1]; ps[PS_ELL + 2] = (ST)e[2];
    ps[PS_ELL2] = (ST)(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
  }
}

template <class T, class ST>
RRT_HD void pack_header(const float* cam, const float* light, ST* hdr) {
  T L[3] = {T(light[0]), T(light[1]), T(light[2])};
  T inv = T(1) / sqrt(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]);
  for (int q = 0; q < HDR_STRIDE; ++q) hdr[q] = ST(0);
  for (int q = 0; q < 3; ++q) {
    hdr[HDR_LHAT + q] = (ST)(L[q] * inv);
    hdr[HDR_LOOK + q] = (ST)cam[12 + q];
    hdr[HDR_CAMO + q] = (ST)cam[4 * q + 3];
    for (int c = 0; c < 3; ++c) hdr[HDR_CAM + 3 * q + c] = (ST)cam[4 * q + c];
  }
  hdr[HDR_INVLN] = (ST)inv;
  hdr[HDR_LLOOK] = (ST)((L[0] * T(cam[12]) + L[1] * T(cam[13]) + L[2] * T(cam[14])) * inv);
}

// Chain one accumulator row back to the raw parameters of its shape:
//   A' = A C, o' = A oc + b  =>  g_A = G_rd C^T + G_o oc^T, g_b = G_o
// Adds into gw[12] (row-major [g_A | g_b]).
template <class T, class AT, class ST>
RRT_HD void finalize_w2o(const AT* acc, const ST* hdr, T gw[12]) {
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) {
      T v = T(acc[ACC_RD + 3 * r]) * T(hdr[HDR_CAM + 3 * c]) +
            T(acc[ACC_RD + 3 * r + 1]) * T(hdr[HDR_CAM + 3 * c + 1]) +
            T(acc[ACC_RD + 3 * r + 2]) * T(hdr[HDR_CAM + 3 * c + 2]);
      v += T(acc[ACC_O + r]) * T(hdr[HDR_CAMO + c]);
      gw[4 * r + c] += v;
    }
    gw[4 * r + 3] += T(acc[ACC_O + r]);
  }
}

// colI = color * I  =>  g_color = I * G_colI ; adds into gm[7].
template <class T, class AT>
RRT_HD void finalize_material(const AT* acc, const float* light, T gm[7]) {
  for (int q = 0; q < 3; ++q) gm[q] += T(light[3 + q]) * T(acc[ACC_COLI + q]);
  gm[3] += T(acc[ACC_KS]);
  gm[4] += T(acc[ACC_KD]);
  gm[5] += T(acc[ACC_KA]);
  gm[6] += T(acc[ACC_SH]);
}

// Light: g_I = sum_s color_s * G_colI_s ; g_Lhat = sum_s G_Lhat_s (both added
// into gl[6], direction part still w.r.t. Lhat); finish with finalize_light_dir.
template <class T, class AT>
RRT_HD void finalize_light_accum(const AT* acc, const float* mat, T gl[6]) {
  for (int q = 0; q < 3; ++q) {
    gl[q] += T(acc[ACC_LHAT + q]);
    gl[3 + q] += T(mat[q]) * T(acc[ACC_COLI + q]);
  }
}

// Lhat = L/|L|  =>  g_L = (g_Lhat - Lhat (Lhat . g_Lhat)) / |L|
template <class T, class ST>
RRT_HD void finalize_light_dir(const ST* hdr, T gl[6]) {
  T Lh[3] = {T(hdr[HDR_LHAT]), T(hdr[HDR_LHAT + 1]), T(hdr[HDR_LHAT + 2])};
  T dot = Lh[0] * gl[0] + Lh[1] * gl[1] + Lh[2] * gl[2];
  for (int q = 0; q < 3; ++q) gl[q] = (gl[q] - Lh[q] * dot) * T(hdr[HDR_INVLN]);
}

}  // namespace rrt
