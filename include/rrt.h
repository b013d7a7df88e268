/*
 * rrt.h -- C ABI of librrt_hip.so, the MI355X (gfx950) render hot path.
 *
 * The reference (lebek/reversible-raytracer) has no FFI / plugin boundary: it is
 * pure Python + Theano.  Its de-facto operator boundary for this path is
 *
 *   Scene.build(antialias_samples=4)  -> image expression   (scene.py:18-52)
 *   T.grad(loss, params)              -> parameter gradients (optimize.py:25)
 *
 * i.e. a pure function (per-shape w2o 4x4, per-shape Material, lights[0],
 * Camera) -> image[x_dims, y_dims, 3] and its vector-Jacobian product.  The
 * entry points below are exactly that pair, with plain pointers and sizes so a
 * ctypes / cffi / CPython stub can bind them (INTEGRATION.md shows the
 * reference-side binding).  No torch / C++ types cross this boundary.
 *
 * Conventions
 *   - every `const float*` / `float*` / `void* workspace` is a DEVICE pointer
 *     owned by the caller; the library allocates nothing and keeps no state
 *   - all calls are asynchronous on `hip_stream` (a hipStream_t, NULL = default
 *     stream); the library never synchronises the device
 *   - return value: RRT_OK (0) or a negative rrt_status; never throws / exits
 *   - thread-safe and re-entrant (no globals besides the last HIP error code)
 *   - outputs are OVERWRITTEN (no pre-zeroing needed by the caller)
 */
#ifndef RRT_H_
#define RRT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRT_VERSION 100 /* 0.1.0 */

typedef enum rrt_status {
  RRT_OK = 0,
  RRT_ERR_NULL = -1,       /* required pointer is NULL */
  RRT_ERR_SHAPE = -2,      /* non-positive / inconsistent dimension */
  RRT_ERR_TOO_MANY = -3,   /* S > RRT_MAX_SHAPES */
  RRT_ERR_WORKSPACE = -4,  /* workspace too small / NULL / misaligned */
  RRT_ERR_ENUM = -5,       /* unknown shader / jitter mode / shape kind */
  RRT_ERR_HIP = -6         /* a HIP runtime call failed: rrt_last_hip_error() */
} rrt_status;

#define RRT_MAX_SHAPES 1024

/* Shape kinds (reference shape.py:72 Sphere, shape.py:16 Square). */
#define RRT_KIND_SPHERE 0
#define RRT_KIND_SQUARE 1

/* Shaders (reference shader.py:23 PhongShader, shader.py:9 DepthMapShader). */
#define RRT_SHADER_PHONG 0
#define RRT_SHADER_DEPTH 1

/* Anti-alias jitter source for scene.py:24-32.
 *   CENTER   U = 0.5 for every (pixel, sample)            (deterministic)
 *   EXPLICIT caller supplies sampleDist_x/_y               (parity runs); values must lie in
 *            [0, 1] as the reference's (U + sample) / A do -- the tile cull bounds a tile's
 *            rays with that reach (pass RRT_FLAG_NO_CULL for anything else)
 *   HASH     U = counter hash of (seed, i, j, sample, axis) (reproducible
 *            stand-in for the reference's unseeded np.random.random) */
#define RRT_JITTER_CENTER 0
#define RRT_JITTER_EXPLICIT 1
#define RRT_JITTER_HASH 2

/* rrt_view.flags */
#define RRT_FLAG_NO_CULL 1u      /* brute force: test every shape at every ray */
#define RRT_FLAG_REUSE_SETUP 2u  /* the packed scene, tile lists and work-queue
                                    batches left in `workspace` by an earlier
                                    render call with the SAME scene, camera and
                                    view are still valid: skip re-packing and
                                    re-dealing (backward right after its forward;
                                    repeated forwards of an unchanged scene) */

#define RRT_FLAG_SHADOWS 8u       /* EXTENSION, default off, Phong shader only: hard shadows after the reference's dormant
                                    Sphere.shadow (shape.py:85-97; its only call site, scene.py:41-45, is commented
                                    out, so no render of the reference shows them).  The point a ray hits is black
                                    when another Sphere lies between it and lights[0]; no gradient flows through
                                    the (piecewise constant) shadow mask.  Squares receive shadows but cast none. */

#define RRT_FLAG_SKIP_FINALIZE 4u /* rrt_render_bwd, measurement only: run the adjoint kernel but not the
                                    final chain to w2o/material/light (g_* are left untouched and the
                                    accumulators un-rewound: a normal rrt_render_bwd WITHOUT
                                    RRT_FLAG_REUSE_SETUP must follow before gradients are used again) */

/* Scene parameters: everything Scene.build() reads from its shapes and
 * lights[0] (scene.py:12-16).  B >= 1 scenes are rendered by one call; an
 * array with batch stride 0 is shared by all B scenes (its gradient is then
 * summed over the batch). */
typedef struct rrt_scene {
  int32_t S; /* shapes per scene, 0..RRT_MAX_SHAPES (list order = composite order, scene.py:38) */
  int32_t B; /* scenes in the batch */
  const float* w2o;        /* [B|1][S][12]  rows 0..2 of shape.w2o.m, row-major [A|b] (shape.py:75) */
  int64_t w2o_bstride;     /* floats between scenes: 0 or S*12 */
  const float* material;   /* [B|1][S][7]   r,g,b,ks,kd,ka,shininess (scene.py:90-101) */
  int64_t material_bstride;/* 0 or S*7 */
  const float* light;      /* [6]           lights[0]: direction xyz (un-normalised), intensity rgb (scene.py:78-86) */
  const uint8_t* kind;     /* [S]           RRT_KIND_*; NULL = all spheres */
} rrt_scene;

/* View: Camera + shader + anti-alias options + the output window. */
typedef struct rrt_view {
  int32_t x_dims, y_dims;  /* Camera(x_dims, y_dims) (scene.py:56) */
  int32_t antialias_samples; /* Scene.build(antialias_samples) (scene.py:18) */
  int32_t row0, rows;      /* output window: rows [row0,row0+rows) of axis 0 of the image (multi-GPU row bands) */
  int32_t shader_id;       /* RRT_SHADER_* */
  int32_t specular;        /* Phong: 1 = root shader.py:45 (ks term), 0 = orbit_experiments/shader.py:45,48 */
  int32_t transpose;       /* 1 = root Camera (one Transform.__call__ => image[a,b] shows ray[b,a], transform.py:46);
                              0 = orbit Camera (camera o2w + shape w2o => two transposes cancel) */
  int32_t jitter_mode;     /* RRT_JITTER_* */
  uint32_t seed;           /* RRT_JITTER_HASH */
  uint32_t flags;          /* RRT_FLAG_* */
  float max_depth;         /* DepthMapShader(maxDepth) (shader.py:11) */
  const float* cam;        /* [B|1][16]: rows 0..2 of camera o2w.m row-major (12 floats; identity for the root Camera),
                              look_at xyz (scene.py:57 / orbit scene.py:57), 1 pad */
  int64_t cam_bstride;     /* 0 or 16 */
  const float* jitter;     /* RRT_JITTER_EXPLICIT: [y_dims][x_dims][A][2] = sampleDist_x, sampleDist_y as handed to
                              Camera.make_rays (scene.py:31-32, i.e. (U + sample) / A); else NULL */
} rrt_view;

/* Image geometry.  The image is (x_dims, y_dims, 3) when transpose=1 and
 * (y_dims, x_dims, 3) when transpose=0 (identical for the square cameras of
 * every reference scene); the window is image[row0:row0+rows].  Output /
 * grad_image layout: [B][rows][n_cols][3] float32, channel-last, contiguous. */
int rrt_image_cols(const rrt_view* v);      /* n_cols, or negative rrt_status */
int rrt_image_rows_full(const rrt_view* v); /* full extent of axis 0 */

int rrt_version(void);
const char* rrt_strerror(int status);
/* hipError_t of the most recent failing HIP call on this thread (0 if none). */
int rrt_last_hip_error(void);

/* Bytes of device scratch either render call needs (16-byte aligned base, no
 * initialisation required).  Holds the packed per-(scene,shape) parameters,
 * per-scene headers, the float32 gradient accumulators (up to four copies), the per-tile candidate
 * masks and tile lists, and the work-queue batches of the last render kernel
 * (about 24 B per 64xP-pixel tile).  Returns 0 on invalid arguments. */
size_t rrt_workspace_bytes(const rrt_scene* s, const rrt_view* v);

/* Forward: replaces Scene.build() + theano.function([], image)()
 * (scene.py:18-52, match_mirror.py:36-39). */
int rrt_render_fwd(const rrt_scene* s, const rrt_view* v, float* image_out,
                   void* workspace, size_t workspace_bytes, void* hip_stream);

/* Backward: replaces T.grad(loss, params) (optimize.py:25) given
 * grad_image = dLoss/dimage over the same window.
 *   g_w2o      [B|1][S][12]  (same batching as s->w2o)
 *   g_material [B|1][S][7]   (same batching as s->material; the shininess
 *                             slot is d/dshininess with lanes of non-positive
 *                             pow base contributing 0 -- an extension, Theano
 *                             would give NaN there)
 *   g_light    [6]           (direction: through normed_dir(); intensity)
 * Any of the three may be NULL (not wanted). */
int rrt_render_bwd(const rrt_scene* s, const rrt_view* v, const float* grad_image,
                   float* g_w2o, float* g_material, float* g_light,
                   void* workspace, size_t workspace_bytes, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* RRT_H_ */
