"""CPU oracle for the render hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``reversible-raytracer_amd/``)
never does: it runs the HIP kernels or fails loudly.

What this is
------------
An op-for-op, array-at-a-time restatement (torch-CPU tensors, float64 or
float32) of the reference's Theano graph for

    ray generation -> ray transform -> ray/sphere + ray/square intersection
    -> Phong / depth shading -> painter's depth composite -> AA average

with every quirk of the reference kept (list below).  The backward pass is
torch autograd over this graph, which is what ``T.grad`` (reference
``optimize.py:25``) does to the Theano graph; masked lanes are sanitised
before ``sqrt`` so that "miss => exactly zero gradient" (the only finite
reading of the reference under ``mode=FAST_RUN``, see SURVEY.md section 8 note).

Pinning status
--------------
The reference (Python 2 + Theano) cannot be imported or run in this
environment (ordinary ModuleNotFoundError; nothing was refused).  This oracle
is pinned by the artifacts the reference's own Theano path produced, committed
under ``tests/golden/``:

* the three known-answer tests of ``test/test_transform.py:9-41``,
* ``orbit_experiments/orbit_dataset.npz[99]`` + ``orbit_target.npz`` (render of
  the ``planet_orbit.py:21-39`` scene, uint8, unseeded AA jitter),
* ``output/0.jpg`` (initial render of ``match_mirror.py:16-33``, JPEG).

Those pin the *forward* render (loosely: 8-bit, unseeded jitter).  Nothing in
the reference pins *gradients*: gradient parity is "pinned" only by this
float64 restatement cross-checked with central finite differences
(``finite_difference_grad`` below) -- say so wherever "grad-check vs Theano" is
claimed.

Quirks reproduced (reference file:line)
---------------------------------------
* ray grid rows are x descending, columns y ascending; jitter is added AFTER
  normalisation to components 0 and 1 (scene.py:66-74)
* every ``Transform.__call__`` spatially transposes the ray grid
  (transform.py:46 ``.T``): root scenes see one transpose, orbit scenes
  (camera o2w + shape w2o) two
* spheres have no ``t > 0`` test (shape.py:109-126); squares do (shape.py:28)
* object-space sphere normals, never mapped back to world (shape.py:128-138)
* unclamped diffuse term, ``pow`` of a possibly negative base, only
  ``lights[0]`` (shader.py:28-53); orbit shader drops the specular term
  (orbit_experiments/shader.py:45,48)
* depth shader: no clip, no inf mask (shader.py:14-20)
* strict ``<`` composite in list order, first shape wins ties (scene.py:46-47)
"""
from __future__ import annotations

import numpy as np
import torch

SPHERE = 0
SQUARE = 1

INF = float("inf")


# --------------------------------------------------------------------------
# transform.py
# --------------------------------------------------------------------------
def identity(dtype=torch.float64):
    """transform.py:56-58"""
    return torch.eye(4, dtype=dtype), torch.eye(4, dtype=dtype)


def _diag_or_col(x, dtype, col):
    """4x4 matrix that is the identity with x written on the diagonal
    (col=False) or in the last column (col=True); functional form of the
    reference's ``T.set_subtensor`` chains so autograd sees x."""
    z = torch.zeros((), dtype=dtype)
    o = torch.ones((), dtype=dtype)
    if col:
        rows = [[o, z, z, x[0]], [z, o, z, x[1]], [z, z, o, x[2]], [z, z, z, o]]
    else:
        rows = [[x[0], z, z, z], [z, x[1], z, z], [z, z, x[2], z], [z, z, z, o]]
    return torch.stack([torch.stack(r) for r in rows])


def translate(x, dtype=torch.float64):
    """transform.py:60-75 -> (m, mInv)"""
    x = torch.as_tensor(x, dtype=dtype)
    return _diag_or_col(x, dtype, True), _diag_or_col(-x, dtype, True)


def scale(x, dtype=torch.float64):
    """transform.py:78-93 -> (m, mInv); inverse uses 1./x"""
    x = torch.as_tensor(x, dtype=dtype)
    return _diag_or_col(x, dtype, False), _diag_or_col(1.0 / x, dtype, False)


def rotate(angle, axis, dtype=torch.float64):
    """transform.py:95-122 -> (m, m.T); angle in DEGREES, axis assumed unit."""
    angle = torch.as_tensor(angle, dtype=dtype)
    a = torch.as_tensor(axis, dtype=dtype)
    radians = angle * np.pi / 180.0
    s = torch.sin(radians)
    c = torch.cos(radians)
    rows = [
        [a[0] * a[0] + (1.0 - a[0] * a[0]) * c,
         a[0] * a[1] * (1.0 - c) - a[2] * s,
         a[0] * a[2] * (1.0 - c) + a[1] * s,
         torch.zeros((), dtype=dtype)],
        [a[0] * a[1] * (1.0 - c) + a[2] * s,
         a[1] * a[1] + (1.0 - a[1] * a[1]) * c,
         a[1] * a[2] * (1.0 - c) - a[0] * s,
         torch.zeros((), dtype=dtype)],
        [a[0] * a[2] * (1.0 - c) - a[1] * s,
         a[1] * a[2] * (1.0 - c) + a[0] * s,
         a[2] * a[2] + (1.0 - a[2] * a[2]) * c,
         torch.zeros((), dtype=dtype)],
        [torch.zeros((), dtype=dtype), torch.zeros((), dtype=dtype),
         torch.zeros((), dtype=dtype), torch.ones((), dtype=dtype)],
    ]
    m = torch.stack([torch.stack(r) for r in rows])
    return m, m.T


def compose(t1, t2):
    """transform.py:35-38  (m1,m1Inv)*(m2,m2Inv)"""
    return t1[0] @ t2[0], t2[1] @ t1[1]


def transform_apply(m, origin, rays):
    """transform.py:40-47.  rays [P,Q,3] -> [Q,P,3] (the ``.T`` reverses ALL
    axes of the [4,P,Q] tensordot result, i.e. a spatial transpose)."""
    o4 = torch.stack([origin[0], origin[1], origin[2],
                      torch.ones((), dtype=m.dtype)])
    new_origin = (m @ o4)[:3]
    r4 = torch.cat([rays, torch.zeros_like(rays)[:, :, :1]], dim=2)
    out = torch.tensordot(m, r4, dims=([1], [2]))      # [4,P,Q]
    out = out.permute(2, 1, 0)[:, :, :3]               # numpy/theano .T
    return new_origin, out


# --------------------------------------------------------------------------
# scene.py : Camera.make_rays
# --------------------------------------------------------------------------
def make_rays(x_dims, y_dims, sampleDist_x=None, sampleDist_y=None,
              dtype=torch.float64, i_range=None, j_range=None):
    """scene.py:61-75.  Rays are computed in float64 NumPy, cast to ``dtype``
    (the reference casts to floatX at :72), THEN jittered (:73-74).

    ``i_range``/``j_range`` = (start, stop) restrict the grid to a
    sub-rectangle (used for row-band / sampled parity at sizes where the full
    grid is too large for an array-at-a-time oracle); the linspaces are always
    those of the full grid.
    """
    lin_i = np.linspace(0.5, -0.5, y_dims)
    lin_j = np.linspace(-0.5, 0.5, x_dims)
    i0, i1 = (0, y_dims) if i_range is None else i_range
    j0, j1 = (0, x_dims) if j_range is None else j_range
    rays = np.dstack(np.meshgrid(lin_i[i0:i1], lin_j[j0:j1], indexing="ij"))
    rays = np.dstack([rays, np.ones([i1 - i0, j1 - j0])])
    rays = rays / np.linalg.norm(rays, axis=2)[:, :, None]
    rays = torch.from_numpy(np.ascontiguousarray(rays)).to(dtype)
    if sampleDist_x is not None:
        sdx = torch.as_tensor(sampleDist_x, dtype=dtype)[i0:i1, j0:j1]
        rays[:, :, 0] = rays[:, :, 0] + sdx / x_dims
    if sampleDist_y is not None:
        sdy = torch.as_tensor(sampleDist_y, dtype=dtype)[i0:i1, j0:j1]
        rays[:, :, 1] = rays[:, :, 1] + sdy / y_dims
    return rays


# --------------------------------------------------------------------------
# jitter
# --------------------------------------------------------------------------
def hash_u01(seed, y_dims, x_dims, A):
    """Counter-hash uniform numbers in [0,1), bit-identical to the HIP kernel's
    ``rrt_hash_u01`` (csrc/rrt_device.h).  Not in the reference: it replaces
    the two UNSEEDED ``np.random.random((x,y,A))`` draws of scene.py:24-25 by a
    reproducible stream so that kernel and oracle can share jitter without an
    8 B/ray side input.  Returns U[y_dims, x_dims, A, 2] float32."""
    i = np.arange(y_dims, dtype=np.uint32)[:, None, None, None]
    j = np.arange(x_dims, dtype=np.uint32)[None, :, None, None]
    s = np.arange(A, dtype=np.uint32)[None, None, :, None]
    ax = np.arange(2, dtype=np.uint32)[None, None, None, :]
    with np.errstate(over="ignore"):
        idx = ((i * np.uint32(x_dims) + j) * np.uint32(A) + s) * np.uint32(2) + ax
        h = idx ^ (np.uint32(seed) * np.uint32(0x9E3779B9))
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13)
        h *= np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def sample_dist_from_uniform(U):
    """scene.py:31-32: sampleDist = (U[:,:,s] + s) / A, same stratum index for
    x and y.  U [y,x,A,2] -> sampleDist [y,x,A,2] (float32 arithmetic, as the
    reference's floatX arrays)."""
    U = np.asarray(U, dtype=np.float32)
    A = U.shape[2]
    s = np.arange(A, dtype=np.float32)[None, None, :, None]
    return ((U + s) / np.float32(A)).astype(np.float32)


def center_sample_dist(y_dims, x_dims, A):
    """Jitter fixed at its mean: U = 0.5."""
    return sample_dist_from_uniform(np.full((y_dims, x_dims, A, 2), 0.5, np.float32))


# --------------------------------------------------------------------------
# shape.py
# --------------------------------------------------------------------------
def sphere_hit(rays, origin):
    """shape.py:78-83 (discriminant)."""
    pnorm = torch.dot(origin, origin)
    vnorm = torch.sum(rays * rays, dim=2)
    pdotv = torch.tensordot(rays, origin, dims=1)
    return pdotv * pdotv - vnorm * (pnorm - 1)


def sphere_distance(rays, origin):
    """shape.py:109-126.  +inf where determinant <= 0 or NaN.  No t>0 test.

    ``sqrt`` sees a sanitised argument on masked lanes so that autograd yields
    exact zeros there (FAST_RUN's switch-sinking rewrites, SURVEY 8.A2 note).
    """
    pdotv = torch.tensordot(rays, origin, dims=1)
    vnorm = torch.sum(rays * rays, dim=2)
    det = sphere_hit(rays, origin)
    bad = (det <= 0) | torch.isnan(det)
    safe = torch.where(bad, torch.ones_like(det), det)
    root = torch.sqrt(safe)
    distance1 = (-pdotv - root) / vnorm
    distance2 = (-pdotv + root) / vnorm
    distance = torch.minimum(distance1, distance2)
    return torch.where(bad, torch.full_like(det, INF), distance)


def sphere_normals(rays, origin):
    """shape.py:128-138 (object-space normals, 'need to fix')."""
    distance = sphere_distance(rays, origin)
    distance = torch.where(torch.isinf(distance), torch.zeros_like(distance), distance)
    projections = origin + distance[:, :, None] * rays
    return projections / torch.sqrt(torch.sum(projections ** 2, dim=2))[:, :, None]


def square_hit(rays, origin):
    """shape.py:25-40 -> (mask, ts)."""
    rz = rays[:, :, 2]
    not_parallel = rz != 0
    safe_rz = torch.where(not_parallel, rz, torch.ones_like(rz))
    ts = -origin[2] / safe_rz
    positive_t = ts > 0
    inter = origin + ts[:, :, None] * rays
    in_x = (inter[:, :, 0] > -0.5) & (inter[:, :, 0] < 0.5)
    in_y = (inter[:, :, 1] > -0.5) & (inter[:, :, 1] < 0.5)
    mask = in_x & in_y & positive_t & not_parallel
    ts = torch.where(mask, ts, torch.full_like(ts, INF))
    return mask, ts


def square_normals(rays, origin):
    """shape.py:52-69."""
    mask, _ = square_hit(rays, origin)
    sign = 1.0 if bool(origin[2] > 0) else -1.0
    n = torch.zeros_like(rays)
    n[:, :, 2] = sign
    return n * mask[:, :, None].to(rays.dtype)


def shape_distance(kind, rays, origin):
    if kind == SPHERE:
        return sphere_distance(rays, origin)
    return square_hit(rays, origin)[1]


def shape_normals(kind, rays, origin):
    if kind == SPHERE:
        return sphere_normals(rays, origin)
    return square_normals(rays, origin)


# --------------------------------------------------------------------------
# shader.py
# --------------------------------------------------------------------------
def normed_dir(direction):
    """scene.py:83-86"""
    d = direction
    return d / torch.sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2)


def phong_shade(kind, rays, origin, material, light, look_at, specular=True, shininess_grad="theano"):
    """shader.py:28-53 (specular=False: orbit_experiments/shader.py:45,48).
    material = (r,g,b,ks,kd,ka,shininess); light = (dir xyz, intensity rgb).

    shininess_grad: what d/dshininess means.  "theano" = the literal chain rule
    of ``base ** shininess`` (``log(base) * pow``: NaN as soon as one lane has a
    negative base -- what T.grad would hand back).  "masked" = the product's
    documented extension (include/rrt.h): lanes with a non-positive base
    contribute 0 to d/dshininess; values and every other gradient are
    unchanged."""
    color, ks, kd, ka, sh = material[0:3], material[3], material[4], material[5], material[6]
    L = normed_dir(light[0:3])
    intensity = light[3:6]
    normals = shape_normals(kind, rays, origin)
    ndotl = torch.tensordot(normals, -L, dims=1)
    diffuse = kd * ndotl
    phong = ka + diffuse
    if specular:
        rm = 2.0 * ndotl[:, :, None] * normals + L
        base = torch.tensordot(rm, look_at, dims=1)
        if shininess_grad == "masked":
            pos = base > 0
            safe = torch.where(pos, base, torch.ones_like(base))
            powv = torch.where(pos, torch.exp(sh * torch.log(safe)), torch.pow(base, sh.detach()))
        else:
            powv = torch.pow(base, sh)
        phong = phong + ks * powv
    colorized = phong[:, :, None] * color[None, None, :] * intensity[None, None, :]
    clipped = torch.clamp(colorized, 0, 1)
    distances = shape_distance(kind, rays, origin)
    return torch.where(torch.isinf(distances)[:, :, None],
                       torch.zeros_like(clipped), clipped)


def depth_shade(kind, rays, origin, max_depth):
    """shader.py:14-20: no clip, no inf mask."""
    distance = shape_distance(kind, rays, origin)
    scaled = distance / max_depth
    return (1 - scaled)[:, :, None] * torch.ones(3, dtype=rays.dtype)


# --------------------------------------------------------------------------
# scene.py : Scene.build
# --------------------------------------------------------------------------
def render(w2o, material, light, kinds=None, *, x_dims, y_dims,
           antialias_samples=4, shader="phong", specular=True, max_depth=1.0,
           cam_o2w=None, look_at=(0.0, 0.0, 1.0), sample_dist=None,
           row0=0, rows=None, dtype=torch.float64, shininess_grad="theano", shadows=False):
    """scene.py:18-52 (``Scene.build``), orbit camera orbit_experiments/scene.py:56-80.

    shadows=True is an EXTENSION with no reference behaviour to pin it (parity unpinned): the hard
    shadows of the dormant ``Sphere.shadow`` (shape.py:85-97), whose only call site is commented out
    (scene.py:41-45) and whose helper ``surface_pts`` raises NameError (shape.py:106).  The reading
    implemented here and in the kernels: the point a ray hits on shape k, carried into the object space
    of every OTHER Sphere k2 (y = o'_k2 + t r'_k2), is in shadow iff the ray from y towards the light
    (direction w2o_k2 . (-Lhat)) enters k2's unit sphere ahead of it -- discriminant > 0 and the nearer
    root >= 0, i.e. Sphere.shadow's ``-x - sqrt(decider) >= 0`` generalised to a non-unit direction;
    a shadowed Phong shading becomes (0,0,0) (the commented-out ``broadcasted_switch``).  Squares
    receive shadows but cast none (the reference defines no Square.shadow); Phong shader only.

    w2o        [S,4,4] (or [S,3,4]) world->object matrices (``shape.w2o.m``)
    material   [S,7]   r,g,b,ks,kd,ka,shininess
    light      [6]     direction xyz (un-normalised), intensity rgb (lights[0])
    kinds      list of SPHERE/SQUARE (default all spheres)
    cam_o2w    None (root Camera: origin 0, no transform) or a 4x4 camera
               object->world matrix (orbit Camera: make_rays returns
               o2w(RayField(0, rays)))
    sample_dist [y_dims,x_dims,A,2] the sampleDist_x/_y arrays handed to
               make_rays (see sample_dist_from_uniform); None = centre (U=.5)
    row0, rows restrict the OUTPUT image to rows [row0,row0+rows) (band of
               axis 0 of the (x_dims,y_dims,3) image)

    Returns image [rows, n_cols, 3] where the full image is
    (x_dims, y_dims, 3) as in the reference (square in every shipped scene).
    """
    w2o = torch.as_tensor(w2o).to(dtype) if not torch.is_tensor(w2o) else w2o.to(dtype)
    material = material.to(dtype) if torch.is_tensor(material) else torch.as_tensor(material, dtype=dtype)
    light = light.to(dtype) if torch.is_tensor(light) else torch.as_tensor(light, dtype=dtype)
    look = torch.as_tensor(np.asarray(look_at, dtype=np.float64)).to(dtype)
    S = w2o.shape[0]
    if kinds is None:
        kinds = [SPHERE] * S
    A = int(antialias_samples)
    if sample_dist is None:
        sample_dist = center_sample_dist(y_dims, x_dims, A)
    sample_dist = np.asarray(sample_dist)
    if w2o.shape[-2] == 3:
        last = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=dtype).expand(S, 1, 4)
        w2o = torch.cat([w2o, last], dim=1)
    if cam_o2w is not None:
        cam_o2w = torch.as_tensor(cam_o2w).to(dtype)
        if cam_o2w.shape[0] == 3:
            cam_o2w = torch.cat([cam_o2w, torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=dtype)], 0)

    # number of spatial transposes between the ray grid and the image
    n_transposes = 1 + (1 if cam_o2w is not None else 0)
    if n_transposes % 2 == 1:       # image[a,b] <- ray[i=b, j=a]
        n_rows_full = x_dims
    else:                           # image[a,b] <- ray[i=a, j=b]
        n_rows_full = y_dims
    if rows is None:
        rows = n_rows_full - row0
    band = (row0, row0 + rows)
    if n_transposes % 2 == 1:
        i_range, j_range = None, band
    else:
        i_range, j_range = band, None

    image = None
    for sample in range(A):
        rays = make_rays(x_dims, y_dims,
                         sampleDist_x=sample_dist[:, :, sample, 0],
                         sampleDist_y=sample_dist[:, :, sample, 1],
                         dtype=dtype, i_range=i_range, j_range=j_range)
        origin = torch.zeros(3, dtype=dtype)
        if cam_o2w is not None:
            origin, rays = transform_apply(cam_o2w, origin, rays)
        image_per_sample = None
        min_dists = None
        for k in range(S):
            o_k, r_k = transform_apply(w2o[k], origin, rays)
            dists = shape_distance(kinds[k], r_k, o_k)
            if shader == "phong":
                shadings = phong_shade(kinds[k], r_k, o_k, material[k], light, look,
                                       specular=specular, shininess_grad=shininess_grad)
                if shadows:
                    t_k = torch.where(torch.isinf(dists), torch.zeros_like(dists), dists).detach()
                    Lhat = normed_dir(light[0:3]).detach()
                    shadowed = torch.zeros_like(dists, dtype=torch.bool)
                    for k2 in range(S):
                        if k2 == k or kinds[k2] != SPHERE:
                            continue
                        o_2, r_2 = transform_apply(w2o[k2].detach(), origin, rays)
                        y = o_2 + t_k[:, :, None] * r_2
                        ell = w2o[k2][:3, :3].detach() @ (-Lhat)
                        yl = torch.tensordot(y, ell, dims=1)
                        disc = yl * yl - torch.dot(ell, ell) * (torch.sum(y * y, dim=2) - 1)
                        shadowed = shadowed | ((disc > 0) & (yl < 0) & (torch.sum(y * y, dim=2) >= 1))
                    shadings = torch.where(shadowed[:, :, None], torch.zeros_like(shadings), shadings)
            else:
                shadings = depth_shade(kinds[k], r_k, o_k, max_depth)
            if image_per_sample is None:
                image_per_sample = torch.zeros(dists.shape + (3,), dtype=dtype)
                min_dists = torch.full(dists.shape, INF, dtype=dtype)
            sel = dists < min_dists
            image_per_sample = torch.where(sel[:, :, None], shadings, image_per_sample)
            min_dists = torch.where(sel, dists, min_dists)
        if image_per_sample is None:    # S == 0
            p, q = rays.shape[1], rays.shape[0]
            image_per_sample = torch.zeros(p, q, 3, dtype=dtype)
        image = image_per_sample if image is None else image + image_per_sample
    return image / A


def hit_ids(w2o, kinds=None, *, x_dims, y_dims, antialias_samples=1,
            cam_o2w=None, sample_dist=None, row0=0, rows=None, dtype=torch.float64, with_margin=False):
    """Index of the shape that wins the composite per (sample, pixel), -1 on
    miss.  Test helper: lets parity checks separate pixels whose winner differs
    between float32 and float64 (silhouettes) from genuine errors.

    ``with_margin=True`` also returns, per ray, how close the ray is to one of the
    DISCRETE decisions of the render, as a relative quantity:
      * min over the spheres of |discriminant| / (p.v)^2 (shape.py:78-83's quantity):
        how close the ray is to grazing a sphere.  d colour / d parameter carries
        1 / sqrt(discriminant) (shape.py:109-126's sqrt), so rays with a tiny margin
        have unbounded weight in a gradient and a float32 evaluation of them is
        ill-conditioned whoever does it;
      * min over the squares of the distance of the ray's plane point to an edge
        line, | |x| - 0.5 | / 0.5 and the same in y (shape.py:33-38's strict tests);
      * the relative gap between the two nearest hits (scene.py:46's strict '<').
    A float32 evaluation can take the other branch on such a ray."""
    w2o = torch.as_tensor(w2o).to(dtype)
    S = w2o.shape[0]
    if kinds is None:
        kinds = [SPHERE] * S
    A = int(antialias_samples)
    if sample_dist is None:
        sample_dist = center_sample_dist(y_dims, x_dims, A)
    if w2o.shape[-2] == 3:
        last = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=dtype).expand(S, 1, 4)
        w2o = torch.cat([w2o, last], dim=1)
    n_transposes = 1 + (1 if cam_o2w is not None else 0)
    n_rows_full = x_dims if n_transposes % 2 == 1 else y_dims
    if rows is None:
        rows = n_rows_full - row0
    band = (row0, row0 + rows)
    i_range, j_range = (None, band) if n_transposes % 2 == 1 else (band, None)
    out, margins = [], []
    with torch.no_grad():
        for sample in range(A):
            rays = make_rays(x_dims, y_dims, sample_dist[:, :, sample, 0],
                             sample_dist[:, :, sample, 1], dtype=dtype,
                             i_range=i_range, j_range=j_range)
            origin = torch.zeros(3, dtype=dtype)
            if cam_o2w is not None:
                cam = torch.as_tensor(cam_o2w).to(dtype)
                origin, rays = transform_apply(cam, origin, rays)
            ids = None
            min_dists = None
            margin = None
            for k in range(S):
                o_k, r_k = transform_apply(w2o[k], origin, rays)
                dists = shape_distance(kinds[k], r_k, o_k)
                if ids is None:
                    ids = torch.full(dists.shape, -1, dtype=torch.int64)
                    min_dists = torch.full(dists.shape, INF, dtype=dtype)
                    margin = torch.full(dists.shape, INF, dtype=dtype)
                if with_margin:
                    # the gap to the nearest hit so far (depth ties), before this shape is merged in
                    both = torch.isfinite(dists) & torch.isfinite(min_dists)
                    gap = (dists - min_dists).abs() / torch.maximum(dists.abs(), min_dists.abs()).clamp_min(1e-300)
                    margin = torch.where(both, torch.minimum(margin, gap), margin)
                sel = dists < min_dists
                ids = torch.where(sel, torch.full_like(ids, k), ids)
                min_dists = torch.where(sel, dists, min_dists)
                if with_margin and kinds[k] == SPHERE:
                    pdotv = torch.tensordot(r_k, o_k, dims=1)
                    margin = torch.minimum(margin, sphere_hit(r_k, o_k).abs() / (pdotv * pdotv).clamp_min(1e-300))
                elif with_margin:
                    rz = r_k[..., 2]
                    ok = rz != 0
                    tt = -o_k[2] / torch.where(ok, rz, torch.ones_like(rz))
                    px, py = o_k[0] + tt * r_k[..., 0], o_k[1] + tt * r_k[..., 1]
                    edge = torch.minimum((px.abs() - 0.5).abs(), (py.abs() - 0.5).abs()) / 0.5
                    # (only rays that reach the plane in front of them anywhere near the square)
                    near_sq = ok & (tt > 0) & (px.abs() < 0.75) & (py.abs() < 0.75)
                    margin = torch.where(near_sq, torch.minimum(margin, edge), margin)
            out.append(ids)
            margins.append(margin)
    if with_margin:
        return torch.stack(out), torch.stack(margins)
    return torch.stack(out)


# --------------------------------------------------------------------------
# gradients
# --------------------------------------------------------------------------
def render_and_grads(w2o, material, light, kinds=None, *, grad_image=None,
                     loss_fn=None, dtype=torch.float64, **kw):
    """Forward + reverse-mode (what ``T.grad`` at optimize.py:25 does).

    Either ``grad_image`` (upstream dL/dimage, same shape as the image) or
    ``loss_fn(image) -> scalar``.  Returns (image, g_w2o[S,3,4], g_material[S,7],
    g_light[6]) as float64/float32 torch tensors (detached)."""
    w2o = torch.as_tensor(w2o).to(dtype).clone().requires_grad_(True)
    material = torch.as_tensor(material).to(dtype).clone().requires_grad_(True)
    light = torch.as_tensor(light).to(dtype).clone().requires_grad_(True)
    image = render(w2o, material, light, kinds, dtype=dtype, **kw)
    if loss_fn is not None:
        loss = loss_fn(image)
    else:
        loss = (image * torch.as_tensor(grad_image).to(dtype)).sum()
    if w2o.shape[0] == 0:
        return image.detach(), torch.zeros(0, 3, 4, dtype=dtype), \
            torch.zeros(0, 7, dtype=dtype), torch.zeros(6, dtype=dtype)
    g = torch.autograd.grad(loss, [w2o, material, light], allow_unused=True)
    g = [torch.zeros_like(p) if gi is None else gi for gi, p in zip(g, (w2o, material, light))]
    return image.detach(), g[0][:, :3, :].detach(), g[1].detach(), g[2].detach()


def finite_difference_grad(fn, x, eps=1e-6):
    """Central finite differences of scalar ``fn`` w.r.t. float64 tensor x."""
    x = x.detach().clone().to(torch.float64)
    g = torch.zeros_like(x)
    flat = x.view(-1)
    gf = g.view(-1)
    for n in range(flat.numel()):
        old = flat[n].item()
        flat[n] = old + eps
        fp = float(fn(x))
        flat[n] = old - eps
        fm = float(fn(x))
        flat[n] = old
        gf[n] = (fp - fm) / (2 * eps)
    return g
