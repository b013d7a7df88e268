"""GPU parity: the HIP path (through the C ABI, via rrt.render) against the
float64 CPU oracle on the same seeded inputs, plus size-independent properties
at full BASELINE sizes.

Tolerances (float32 kernel vs float64 oracle), chosen from the float32 noise
floor of the same formulas measured on the CPU harness (test_device_math.py):
  * image: |d| <= IMG_TOL on every pixel except at most MAX_FLIP_FRAC of them
    (silhouette pixels whose hit/miss flips between float32 and float64);
  * parameter gradients, per tensor: max|d| <= GRAD_TIGHT * max|g| (1e-3, SURVEY
    section 7).  A check that misses it may still pass on the float32 noise floor
    of these formulas -- the adjoint divides by sqrt(Delta), which amplifies
    float32 rounding at grazing rays, and the reference itself rounds its rays to
    float32 -- measured at the same point by the oracle run in float32:
    max|d| <= 4 x (float32 oracle's error), and that only up to GRAD_RTOL * max|g|
    (5e-3; 2e-2 for the thin white-noise band) unless a ray is present whose
    float32 arithmetic is undecidable: a hit/miss flip (a pixel of the GPU image
    off by more than IMG_TOL, or a ray whose winner differs between the float32
    and the float64 oracle, ``oracle.hit_ids``), or a ray so close to grazing that
    the float32 ORACLE ITSELF misses that cap on the tensor (round-3 soak of 200
    random scenes: eight such cases, no flip, float32 oracle off by 1.3e-2 ...
    1.7e-1 of max|g|, the GPU by 5e-3 ... 1.6e-2 -- its discriminant is the better
    conditioned one -- and the product's own math in float64 within 5e-6).
    A ray within GRAZING_MARGIN of any discrete decision of the render (sphere
    silhouette, Square edge, depth tie: ``oracle.hit_ids(with_margin=True)``)
    counts as undecidable too.
    With such an undecidable ray present even that bound can fail -- the ray's
    weight 1 / sqrt(discriminant) is unbounded and one of them can outweigh the
    rest of the image (seen: 8.7 x max|g| at 33 x 33 pixels) -- and then the
    masked comparison below is the test;
  * per entry: every gradient entry, small ones included, within
    max(ENTRY_RTOL * |entry|, 4 x the float32 oracle's error on that entry,
    ENTRY_FLOOR * max|g|) -- so an entry above ENTRY_FLOOR/ENTRY_RTOL of the
    tensor's scale is held to a relative error of its own;
  * whenever the float32-floor clause was needed, the comparison is repeated with
    the upstream gradient zeroed on the pixels that have a sample ray within
    GRAZING_MARGIN (relative discriminant, ``oracle.hit_ids(with_margin=True)``)
    of a sphere's silhouette: there GPU and float64 oracle must agree to
    GRAD_TIGHT with no fallback -- i.e. the 1 / sqrt(discriminant) weight of
    a handful of grazing rays is shown to be the whole reason;
  * d/dshininess (material column 6) is compared like every other entry, against
    the oracle's "masked" exponent gradient (the product's documented extension:
    lanes with a non-positive pow base contribute 0; include/rrt.h).
Every check appends (test, tensor, error / scale, flips) to the report printed at
the end of the module (and to $RRT_PARITY_REPORT when set).
"""
import json
import os
import numpy as np
import pytest
import torch

import rrt_oracle as O

pytestmark = pytest.mark.gpu

IMG_TOL = 2e-4
MAX_FLIP_FRAC = 5e-4
GRAD_TIGHT = 1e-3
GRAD_RTOL = 5e-3
ENTRY_RTOL = 5e-3
# (float32 noise on small entries, round-3 soak: 3e-5 and 8.2e-5 of max|g| on entries 1.5e-3 / 3e-4 of the tensor's scale (seeds
# 159, 259); the product's math run in float32 on the CPU shows the same 8e-5, in float64 3e-11)
# a last soak of 500 more renders (seeds 600..849): one entry at 1.03e-4 (seed 729, an entry of 5e-3 of the scale; the product's
# math on the CPU: 1.09e-4 in float32, 7e-12 in float64) -- the floor is 2e-4
ENTRY_FLOOR = 2e-4
GRAZING_MARGIN = 1e-4

_REPORT = []


def _note(what, rel, flips, fallback):
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0]
    _REPORT.append({"test": test, "tensor": what, "rel_err": rel, "flips": flips, "fallback": fallback})


@pytest.fixture(scope="module", autouse=True)
def _parity_report():
    yield
    if _REPORT:
        worst = max(_REPORT, key=lambda r: r["rel_err"])
        print("\nparity report: %d gradient checks, %d needed the float32-floor clause; worst %s" % (
            len(_REPORT), sum(r["fallback"] for r in _REPORT), json.dumps(worst)))
        path = os.environ.get("RRT_PARITY_REPORT")
        if path:
            with open(path, "w") as f:
                json.dump(_REPORT, f, indent=1)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from rrt import _native
    _native.lib()          # must exist on a GPU box: no silent fallback
    return torch.device("cuda:0")


def hip_render(dev, w2o, mat, light, kinds, grad=None, **kw):
    import rrt
    tw = torch.tensor(w2o, device=dev, requires_grad=True)
    tm = torch.tensor(mat, device=dev, requires_grad=True)
    tl = torch.tensor(light, device=dev, requires_grad=True)
    cam = kw.pop("cam", None)
    if cam is not None:
        cam = torch.tensor(cam, device=dev)
    img = rrt.render(tw, tm, tl, kinds, cam, **kw)
    if grad is None:
        return img.detach().cpu().double().numpy()
    g = torch.tensor(np.asarray(grad, dtype=np.float32), device=dev)
    (img * g).sum().backward()
    torch.cuda.synchronize()
    return (img.detach().cpu().double().numpy(), tw.grad.cpu().double().numpy(),
            tm.grad.cpu().double().numpy(), tl.grad.cpu().double().numpy())


def oracle(w2o, mat, light, kinds, grad, *, x_dims, y_dims, antialias_samples, jitter=None, cam=None,
           shader="phong", specular=True, max_depth=1.0, row0=0, rows=None, dtype=torch.float64, want_flips=False,
           shadows=False, want_margin=False):
    sd = None
    if isinstance(jitter, (int, np.integer)):
        sd = O.sample_dist_from_uniform(O.hash_u01(jitter, y_dims, x_dims, antialias_samples))
    elif jitter is not None and not isinstance(jitter, str):
        sd = np.asarray(jitter)
    cam_o2w = None
    if cam is not None:
        cam_o2w = np.eye(4)
        cam_o2w[:3, :4] = np.asarray(cam[:12]).reshape(3, 4)
    kw = dict(x_dims=x_dims, y_dims=y_dims, antialias_samples=antialias_samples, shader=shader, specular=specular,
              max_depth=max_depth, cam_o2w=cam_o2w, sample_dist=sd, row0=row0, rows=rows, shininess_grad="masked",
              shadows=shadows)
    kl = list(kinds) if kinds is not None else None
    if want_margin:                     # per pixel: how close its closest-to-grazing sample ray is to a sphere's silhouette
        hk = {k: kw[k] for k in ("x_dims", "y_dims", "antialias_samples", "cam_o2w", "sample_dist", "row0", "rows")}
        return O.hit_ids(w2o, kl, with_margin=True, **hk)[1].amin(0).numpy()
    img, gw, gm, gl = O.render_and_grads(w2o, mat, light, kl, grad_image=grad, dtype=dtype, **kw)
    if want_flips:
        hk = {k: kw[k] for k in ("x_dims", "y_dims", "antialias_samples", "cam_o2w", "sample_dist", "row0", "rows")}
        ids = [O.hit_ids(w2o, kl, dtype=dt, **hk) for dt in (torch.float32, torch.float64)]
        return int((ids[0] != ids[1]).sum())
    return img.double().numpy(), gw.double().numpy(), gm.double().numpy(), gl.double().numpy()


def assert_image_close(a, b, what=""):
    err = np.abs(a - b).max(-1)
    bad = int((err > IMG_TOL).sum())
    allowed = max(2, int(MAX_FLIP_FRAC * err.size))
    assert bad <= allowed, "%s: %d pixels off by > %g (allowed %d), max %g" % (what, bad, IMG_TOL, allowed, err.max())


def assert_grad_close(a, b, what="", rtol=None):
    """plain max-norm check (structural tests: two GPU runs against each other, sums over a batch)"""
    rtol = GRAD_RTOL if rtol is None else rtol
    scale = max(np.abs(b).max(), 1e-20)
    err = np.abs(a - b).max()
    assert err <= rtol * scale, "%s: max|d|=%g vs scale %g (rel %g)" % (what, err, scale, err / scale)


def assert_grads_vs_oracle(got, ref, f32, image_flips, count_flips, loose=None, count_near=None):
    """got / ref / f32: (g_w2o, g_material[S,7], g_light) of the GPU, the float64 oracle and the
    float32 oracle (lazily: `f32` and `count_flips` are callables, only evaluated when the tight
    bound fails).  See the module docstring for the bounds."""
    cache = {}
    used_fallback = False

    def floor():
        if "f32" not in cache:
            cache["f32"] = f32()
        return cache["f32"]

    def flips():
        if "flips" not in cache:
            cache["flips"] = image_flips + count_flips()
        return cache["flips"]
    for idx, what in enumerate(("g_w2o", "g_material", "g_light")):
        a, b = np.asarray(got[idx]), np.asarray(ref[idx])
        assert np.isfinite(a).all(), what
        if b.size == 0:
            continue
        scale = max(np.abs(b).max(), 1e-20)
        err = np.abs(a - b)
        tol, fallback, undecidable = GRAD_TIGHT * scale, False, False
        if err.max() > tol:
            fl_raw = np.abs(np.asarray(floor()[idx]) - b).max()
            fl = 4.0 * fl_raw
            cap = (loose or GRAD_RTOL) * scale
            # undecidable in float32: a hit/miss flip, the float32 oracle itself beyond the cap on this tensor, or a ray
            # within GRAZING_MARGIN of a discrete decision (sphere silhouette, square edge, depth tie: oracle.hit_ids)
            if "near" not in cache:
                cache["near"] = count_near() if count_near is not None else 0
            undecidable = flips() > 0 or fl_raw > cap or cache["near"] > 0
            tol, fallback = max(tol, fl if undecidable else min(fl, cap)), True
            used_fallback = True
        _note(what, float(err.max() / scale), cache.get("flips", image_flips), fallback)
        if fallback and undecidable and err.max() > tol:
            # A ray whose hit / miss is undecidable in float32 is present, and its weight 1 / sqrt(discriminant) has no
            # bound: one such ray can outweigh the rest of the image (seen: 8.7 x max|g|), in the GPU's float32, in the
            # float32 oracle's, or in both, differently.  No bound on the full sum is meaningful then -- the caller's
            # comparison with the grazing pixels masked (check_scene) is the test, and it has no fallback.
            _note(what + " [undecidable ray: masked comparison decides]", float(err.max() / scale), flips(), True)
            continue
        assert err.max() <= tol, "%s: max|d|=%g vs scale %g (rel %.2e), flips %d" % (what, err.max(), scale, err.max() / scale,
                                                                                    flips())
        if undecidable:
            continue                 # (entries: the masked comparison checks them, with the floor only)
        # per entry (small entries too)
        ent = np.maximum(ENTRY_RTOL * np.abs(b), ENTRY_FLOOR * scale)
        if (err > ent).any():
            fl = np.abs(np.asarray(floor()[idx]) - b)
            ent = np.maximum(ent, 4.0 * fl)
        worst = np.unravel_index(np.argmax(err - ent), err.shape)
        assert (err <= ent).all(), "%s%s: |d|=%g, entry %g, allowed %g (scale %g)" % (what, list(worst), err[worst], b[worst],
                                                                                    ent[worst], scale)
    return used_fallback


def assert_grads_tight(got, ref, tag):
    """1e-3 * max|g| per tensor and max(ENTRY_RTOL |entry|, ENTRY_FLOOR max|g|) per entry, no fallback"""
    for name, a, r in zip(("g_w2o", "g_material", "g_light"), got, ref):
        if np.asarray(r).size == 0:
            continue
        a, r = np.asarray(a), np.asarray(r)
        scale = max(np.abs(r).max(), 1e-20)
        err = np.abs(a - r)
        _note("%s [%s]" % (name, tag), float(err.max() / scale), 0, False)
        assert err.max() <= GRAD_TIGHT * scale, "%s (%s): %.3g vs scale %.3g (%.2e rel)" % (name, tag, err.max(), scale, err.max() / scale)
        ent = np.maximum(ENTRY_RTOL * np.abs(r), ENTRY_FLOOR * scale)
        worst = np.unravel_index(np.argmax(err - ent), err.shape)
        assert (err <= ent).all(), "%s%s (%s): |d|=%g, entry %g, allowed %g (scale %g)" % (name, list(worst), tag, err[worst], r[worst],
                                                                                        ent[worst], scale)


def check_scene(dev, w2o, mat, light, kinds, seed=0, grad_rtol=None, **kw):
    okw = dict(kw)
    okw.pop("cull", None)
    rows = kw.get("rows")
    n_rows = rows if rows is not None else (kw["x_dims"] if kw.get("cam") is None else kw["y_dims"]) - kw.get("row0", 0)
    n_cols = kw["y_dims"] if kw.get("cam") is None else kw["x_dims"]
    G = np.random.RandomState(seed).randn(n_rows, n_cols, 3)
    hi, hw, hm, hl = hip_render(dev, w2o, mat, light, kinds, grad=G, **kw)
    oi, ow, om, ol = oracle(w2o, mat, light, kinds, G, **okw)
    assert_image_close(hi, oi, "image")
    image_flips = int((np.abs(hi - oi).max(-1) > IMG_TOL).sum())
    fell_back = assert_grads_vs_oracle((hw, hm, hl), (ow, om, ol),
                                       lambda: oracle(w2o, mat, light, kinds, G, dtype=torch.float32, **okw)[1:],
                                       image_flips, lambda: oracle(w2o, mat, light, kinds, G, want_flips=True, **okw),
                                       loose=grad_rtol,
                                       count_near=lambda: int((oracle(w2o, mat, light, kinds, None, want_margin=True, **okw) < GRAZING_MARGIN).sum()))
    if fell_back:
        assert_grads_without_grazing_pixels(dev, w2o, mat, light, kinds, G, kw, okw)
    return hi


def assert_grads_without_grazing_pixels(dev, w2o, mat, light, kinds, G, kw, okw):
    """the tight bound was missed on the full upstream gradient: show that the near-grazing rays (module docstring)
    are the whole reason -- with their pixels' upstream gradient zeroed, GPU and oracle must agree to 1e-3, no fallback"""
    well = oracle(w2o, mat, light, kinds, None, want_margin=True, **okw) >= GRAZING_MARGIN
    assert 0 < (~well).sum() < 0.25 * well.size, (~well).sum()
    Gw = G * well[:, :, None]
    _, *h_well = hip_render(dev, w2o, mat, light, kinds, grad=Gw, **kw)
    _, *o_well = oracle(w2o, mat, light, kinds, Gw, **okw)
    assert_grads_tight(h_well, o_well, "grazing pixels masked: %d of %d" % ((~well).sum(), well.size))


# ---------------------------------------------------------------- configs --
def test_c1_plumbing_1sphere_64(dev):
    """BASELINE config 1 geometry: one sphere, 64x64, reference default A=4."""
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.one_sphere()
    check_scene(dev, w2o, mat, light, kinds, x_dims=64, y_dims=64, antialias_samples=4, jitter=11)


def test_c2_1sphere_phong_512(dev):
    """BASELINE config 2: 1 sphere Phong 512x512, fwd + analytic bwd."""
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.one_sphere()
    check_scene(dev, w2o, mat, light, kinds, x_dims=512, y_dims=512, antialias_samples=1)


@pytest.mark.parametrize("with_square", [False, True])
def test_c3_match_mirror(dev, with_square):
    """BASELINE config 3 scene (match_mirror.py:16-29), 256x256 here (the
    oracle's array-at-a-time autograd at 1024^2 is minutes; the 1024^2 case is
    covered by the band test below)."""
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.match_mirror(with_square)
    check_scene(dev, w2o, mat, light, kinds, x_dims=256, y_dims=256, antialias_samples=4, jitter=5)


def test_c3_match_mirror_1024_band(dev):
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.match_mirror(True)
    check_scene(dev, w2o, mat, light, kinds, x_dims=1024, y_dims=1024, antialias_samples=1, row0=400, rows=48)


def test_c3_match_mirror_1024_full(dev):
    """BASELINE config 3 at its stated size: the match_mirror.py scene (2 spheres + Square), 1024 x 1024,
    antialias_samples = 4, ONE launch on the GPU; the oracle covers the whole image in eight 128-row bands
    (its array-at-a-time autograd would need tens of GB at once) whose gradients add up.

    Two upstream gradients, because d colour / d centre carries 1 / sqrt(discriminant) (shape.py:109-126): with 4 M
    rays the closest-to-grazing one has a relative discriminant of ~1e-7 and weighs as much as all other rays
    together, in float32 it is noise (or hits on one side and misses on the other), for the oracle in float32 too.
      (1) white noise with the pixels zeroed whose closest sample ray is within GRAZING_MARGIN (relative
          discriminant) of a sphere's silhouette -- 0.4 % of the pixels: 1e-3 * max|g|, no fallback (measured 1.1e-5; 4.8e-5 with a
          margin of 1e-5, 535 pixels);
      (2) the same noise on all pixels: the float32-floor rule of the other tests."""
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.match_mirror(True)
    n, A, nb = 1024, 4, 8
    G = np.random.RandomState(3).randn(n, n, 3)
    bands = [(b * (n // nb), n // nb) for b in range(nb)]
    kws = [dict(x_dims=n, y_dims=n, antialias_samples=A, jitter=5, row0=r0, rows=nr) for r0, nr in bands]
    margin = np.concatenate([oracle(w2o, mat, light, kinds, None, want_margin=True, **kw) for kw in kws])
    assert margin.shape == (n, n)
    well = margin >= GRAZING_MARGIN
    assert 0 < (~well).sum() < 5e-3 * n * n, (~well).sum()           # measured: 3819 of 1048576
    Gw = G * well[:, :, None]
    Gr = G - Gw
    hi, *h_well = hip_render(dev, w2o, mat, light, kinds, grad=Gw, x_dims=n, y_dims=n, antialias_samples=A, jitter=5)
    _, *h_all = hip_render(dev, w2o, mat, light, kinds, grad=G, x_dims=n, y_dims=n, antialias_samples=A, jitter=5)
    o_well = [np.zeros_like(x) for x in h_well]
    o_rest = [np.zeros_like(x) for x in h_well]
    f_all = [np.zeros_like(x) for x in h_well]
    flips = 0
    for b, ((r0, nr), kw) in enumerate(zip(bands, kws)):
        oi, *ow = oracle(w2o, mat, light, kinds, Gw[r0:r0 + nr], **kw)
        assert_image_close(hi[r0:r0 + nr], oi, "image rows %d.." % r0)
        flips += int((np.abs(hi[r0:r0 + nr] - oi).max(-1) > IMG_TOL).sum())
        _, *orr = oracle(w2o, mat, light, kinds, Gr[r0:r0 + nr], **kw)
        for acc, x in zip(o_well, ow):
            acc += x
        for acc, x in zip(o_rest, orr):
            acc += x
        if b in (2, 5):                                      # float32 floor on two bands (scaled up below): cheap enough
            f = oracle(w2o, mat, light, kinds, G[r0:r0 + nr], dtype=torch.float32, **kw)
            for acc, x, y, z in zip(f_all, f[1:], ow, orr):
                acc += x - y - z
    # (1) hard bound, no fallback
    assert_grads_tight(h_well, o_well, "grazing pixels masked: %d of %d" % ((~well).sum(), well.size))
    # (2) every pixel
    o_all = [x + y for x, y in zip(o_well, o_rest)]
    assert_grads_vs_oracle(tuple(h_all), tuple(o_all), lambda: tuple(o + 4 * f for o, f in zip(o_all, f_all)), flips, lambda: 0)


def test_depth_shader(dev):
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.match_mirror(True)
    check_scene(dev, w2o, mat, light, kinds, x_dims=128, y_dims=128, antialias_samples=2, shader="depth",
                max_depth=20.0)


def test_orbit_camera_no_specular(dev, golden_dir):
    """orbit_experiments scene (planet_orbit.py:21-39): orbit camera (no net
    transpose), no specular term; also checked against the reference's own
    render orbit_dataset.npz[99] (8-bit, unseeded jitter => loose)."""
    import os
    from rrt import synthetic
    fx = np.load(os.path.join(golden_dir, "orbit_sample99.npz"))
    w2o, mat, light, kinds = synthetic.orbit_scene(fx["target"])
    for c, y in enumerate((2.5, -2.5)):
        cam = synthetic.orbit_camera(y)
        img = check_scene(dev, w2o, mat, light, kinds, x_dims=64, y_dims=64, antialias_samples=4,
                          specular=False, cam=cam, jitter="center" if c == 0 else 9)
        if c == 0:
            q = (img * 255).astype(np.uint8).astype(int)
            d = np.abs(q - fx["images"][c].astype(int)).max(-1)
            assert (d <= 2).mean() > 0.98


def test_match_mirror_reference_jpeg(dev, golden_dir):
    """output/0.jpg: the reference's own initial render of match_mirror.py."""
    import os
    from rrt import synthetic
    ref = np.load(os.path.join(golden_dir, "match_mirror_0.npy")).astype(np.float64) / 255.0
    w2o, mat, light, kinds = synthetic.match_mirror(True)
    img = hip_render(dev, w2o, mat, light, kinds, x_dims=128, y_dims=128, antialias_samples=4, jitter="center")
    assert np.abs(img - ref).mean() * 255 <= 3.0


def test_random_spheres_32(dev):
    """32-sphere bench scene at a size the oracle finishes in seconds."""
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.random_spheres(32, seed=0)
    check_scene(dev, w2o, mat, light, kinds, x_dims=192, y_dims=192, antialias_samples=2, jitter=21)


def test_explicit_jitter(dev):
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.random_spheres(5, seed=2)
    n, A = 96, 3
    U = np.random.RandomState(4).rand(n, n, A, 2).astype(np.float32)
    sd = O.sample_dist_from_uniform(U)
    check_scene(dev, w2o, mat, light, kinds, x_dims=n, y_dims=n, antialias_samples=A, jitter=sd)


def test_explicit_jitter_outside_unit_range(dev):
    """sampleDist outside [0, 1] (not something the reference produces): the tile cull's bounds do
    not cover it, so rrt.render switches the cull off for such a tensor -- results still match the
    oracle, including shapes that only the far-displaced rays reach"""
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.random_spheres(6, seed=4)
    n, A = 80, 2
    sd = (np.random.RandomState(7).rand(n, n, A, 2) * 24.0 - 12.0).astype(np.float32)    # up to 12 pixels either way
    check_scene(dev, w2o, mat, light, kinds, x_dims=n, y_dims=n, antialias_samples=A, jitter=sd)


@pytest.mark.parametrize("n,A,specular,tile_rows", [(224, 2, True, 0), (96, 1, False, 0), (300, 1, True, 4), (176, 2, True, 2)])
def test_shadows_extension(dev, n, A, specular, tile_rows, monkeypatch):
    """RRT_FLAG_SHADOWS -- an extension with no reference behaviour behind it (the reference's Sphere.shadow is
    dormant, shape.py:85-97; parity unpinned): the kernels' wave-ballot any-hit walk against the oracle's statement
    of the same reading, image and gradients; multi- and single-candidate tiles, 1-, 2- and 4-row tiles"""
    from rrt import _native
    from test_device_math import _shadow_scene
    w2o, mat, light, kinds = _shadow_scene()
    if tile_rows:
        monkeypatch.setenv("RRT_TUNE_P", str(tile_rows))
        _native.lib().rrt_debug_reload_tuning()
    try:
        img = check_scene(dev, w2o, mat, light, kinds, x_dims=n, y_dims=n, antialias_samples=A, jitter=13,
                          specular=specular, shadows=True)
        lit = hip_render(dev, w2o, mat, light, kinds, x_dims=n, y_dims=n, antialias_samples=A, jitter=13, specular=specular)
    finally:
        if tile_rows:
            monkeypatch.delenv("RRT_TUNE_P")
            _native.lib().rrt_debug_reload_tuning()
    dark = ((img == 0).all(-1) & (lit > 0).any(-1)).sum()
    assert dark > 0.003 * n * n                                    # the shadow is there


def test_ellipsoid_rotate_scale(dev):
    """optimize_brightness.py:28: translate*rotate*scale (general w2o)."""
    O2 = O.compose(O.compose(O.translate((.5, .5, 4.)), O.rotate(90., (0., 0., 1.))), O.scale((1., 2., 1.5)))
    O1 = O.translate((-.5, -.5, 4.))
    w2o = np.stack([O1[1].numpy()[:3], O2[1].numpy()[:3]]).astype(np.float32)
    mat = np.array([[0.2, 0.9, 0.4, 0.3, 0.7, 0.5, 50.], [0.87, 0.1, 0.507, 0.3, 0.9, 0.4, 50.]], np.float32)
    light = np.array([-1., -1., 2., 0.961, 1., 0.87], np.float32)
    check_scene(dev, w2o, mat, light, np.zeros(2, np.uint8), x_dims=128, y_dims=128, antialias_samples=4, jitter=1)


# ------------------------------------------------------------- edge cases --
def test_empty_scene(dev):
    w2o = np.zeros((0, 3, 4), np.float32)
    mat = np.zeros((0, 7), np.float32)
    light = np.array([0, 0, 1, 1, 1, 1], np.float32)
    G = np.ones((40, 40, 3))
    img, gw, gm, gl = hip_render(dev, w2o, mat, light, None, grad=G, x_dims=40, y_dims=40, antialias_samples=2)
    assert img.shape == (40, 40, 3) and (img == 0).all()
    assert gw.shape == (0, 3, 4) and (gl == 0).all()


@pytest.mark.parametrize("n", [1, 3, 63, 65, 130])
def test_ragged_sizes(dev, n):
    """sizes that are not multiples of the 64-wide tile / 4-row wave stride"""
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.match_mirror(True)
    check_scene(dev, w2o, mat, light, kinds, x_dims=n, y_dims=n, antialias_samples=2, jitter=3)


def test_all_miss(dev):
    """sphere behind a far corner: image all zero, gradients exactly zero"""
    from rrt import synthetic
    w2o = synthetic._w2o_translate_scale((50., 50., 1.), .1)[None].astype(np.float32)
    mat = np.array([[.5, .5, .5, .3, .7, .5, 50.]], np.float32)
    light = np.array([0, 0, 1, 1, 1, 1], np.float32)
    G = np.ones((64, 64, 3))
    img, gw, gm, gl = hip_render(dev, w2o, mat, light, None, grad=G, x_dims=64, y_dims=64, antialias_samples=1)
    assert (img == 0).all() and (gw == 0).all() and (gm == 0).all() and (gl == 0).all()


def test_sphere_behind_camera_wins(dev):
    """no t>0 test for spheres (shape.py:109-126): a sphere behind the camera
    yields negative t and wins the depth composite"""
    from rrt import synthetic
    w2o = np.stack([synthetic._w2o_translate_scale((0., 0., 4.), 1.),
                    synthetic._w2o_translate_scale((0., 0., -4.), 1.)]).astype(np.float32)
    mat = np.array([[.9, .1, .1, .3, .7, .5, 50.], [.1, .9, .1, .3, .7, .5, 50.]], np.float32)
    light = np.array([-1, -1, 2, 1, 1, 1], np.float32)
    check_scene(dev, w2o, mat, light, np.zeros(2, np.uint8), x_dims=64, y_dims=64, antialias_samples=1)


def test_tie_first_shape_wins(dev):
    """two identical spheres: strict '<' keeps the first (scene.py:46)"""
    from rrt import synthetic
    one = synthetic._w2o_translate_scale((0., 0., 4.), 1.)
    w2o = np.stack([one, one]).astype(np.float32)
    mat = np.array([[.9, .1, .1, .3, .7, .5, 50.], [.1, .9, .1, .3, .7, .5, 50.]], np.float32)
    light = np.array([-1, -1, 2, 1, 1, 1], np.float32)
    G = np.ones((64, 64, 3))
    img, gw, gm, gl = hip_render(dev, w2o, mat, light, None, grad=G, x_dims=64, y_dims=64, antialias_samples=1)
    assert (gm[1] == 0).all() and (gw[1] == 0).all() and np.abs(gm[0]).max() > 0
    assert img[..., 0].max() > img[..., 1].max()


# ---------------------------------------------------- structural properties --
def test_cull_is_exact(dev):
    """tile culling must not change a single bit (vs brute force)"""
    import rrt
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.random_spheres(32, seed=0)
    a = hip_render(dev, w2o, mat, light, kinds, x_dims=1024, y_dims=1024, antialias_samples=1, cull=True)
    b = hip_render(dev, w2o, mat, light, kinds, x_dims=1024, y_dims=1024, antialias_samples=1, cull=False)
    assert np.array_equal(a, b)


def test_many_shapes_multiword_mask(dev):
    """S > 64: more than one 64-bit candidate word"""
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.random_spheres(150, seed=7)
    check_scene(dev, w2o, mat, light, kinds, x_dims=96, y_dims=96, antialias_samples=1)


def test_bands_tile_the_image_and_gradients_add(dev):
    """full 4096x4096 / 32 spheres: the 8 row bands of the multi-GPU split
    reproduce the single-launch image bit for bit and their parameter
    gradients sum to the single-launch gradient (linearity of the adjoint)."""
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.random_spheres(32, seed=0)
    n = 4096
    rs = np.random.RandomState(0)
    G = rs.randn(n, n, 3).astype(np.float32)
    full = hip_render(dev, w2o, mat, light, kinds, grad=G, x_dims=n, y_dims=n, antialias_samples=1)
    acc = [np.zeros_like(full[1]), np.zeros_like(full[2]), np.zeros_like(full[3])]
    for r in range(8):
        r0, nr = r * 512, 512
        part = hip_render(dev, w2o, mat, light, kinds, grad=G[r0:r0 + nr], x_dims=n, y_dims=n, antialias_samples=1,
                          row0=r0, rows=nr)
        assert np.array_equal(part[0], full[0][r0:r0 + nr])
        for k in range(3):
            acc[k] += part[k + 1]
    for k in range(3):
        scale = np.abs(full[k + 1]).max()
        assert np.abs(acc[k] - full[k + 1]).max() <= 1e-4 * scale
    # coverage sanity of the bench scene (~25-35 %)
    cov = (full[0].max(-1) > 0).mean()
    assert 0.1 < cov < 0.6


def test_full_size_band_vs_oracle(dev):
    """4096x4096 / 32 spheres: a 24-row band against the float64 oracle.

    A thin band with a white-noise upstream gradient is the worst case for the
    float32 noise floor: its parameter gradient is a random-sign sum dominated
    by a handful of grazing pixels (contribution ~ 1/sqrt(Delta)), and the
    reference itself rounds its rays to float32 (scene.py:72), which moves
    Delta by ~1e-7*|o'| there.  Hence the looser gradient tolerance; the
    whole-image tests and the loss-shaped test below keep the tight one."""
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.random_spheres(32, seed=0)
    check_scene(dev, w2o, mat, light, kinds, grad_rtol=2e-2, x_dims=4096, y_dims=4096, antialias_samples=1,
                row0=1900, rows=24)


def test_batch_axis(dev):
    """B scenes in one launch == B separate launches; shared w2o => summed grads"""
    import rrt
    from rrt import synthetic
    rs = np.random.RandomState(3)
    B, n = 5, 64
    base_w, mat, light, kinds = synthetic.orbit_scene()
    w = np.stack([synthetic.orbit_scene((rs.uniform(-9, 9), rs.uniform(-9, 9), 32.))[0] for _ in range(B)])
    cams = np.stack([synthetic.orbit_camera(2.5 if b % 2 == 0 else -2.5) for b in range(B)])
    G = rs.randn(B, n, n, 3).astype(np.float32)
    tw = torch.tensor(w, device=dev, requires_grad=True)
    tm = torch.tensor(mat, device=dev, requires_grad=True)
    tl = torch.tensor(light, device=dev, requires_grad=True)
    img = rrt.render(tw, tm, tl, kinds, torch.tensor(cams, device=dev), x_dims=n, y_dims=n, antialias_samples=2,
                     specular=False, jitter=2)
    (img * torch.tensor(G, device=dev)).sum().backward()
    gm_sum = np.zeros((2, 7))
    gl_sum = np.zeros(6)
    for b in range(B):
        one = hip_render(dev, w[b], mat, light, kinds, grad=G[b], cam=cams[b], x_dims=n, y_dims=n,
                         antialias_samples=2, specular=False, jitter=2)
        assert np.array_equal(one[0], img[b].detach().cpu().double().numpy())
        assert np.abs(one[1] - tw.grad[b].cpu().double().numpy()).max() <= 1e-4 * max(np.abs(one[1]).max(), 1e-20)
        gm_sum += one[2]
        gl_sum += one[3]
    assert np.abs(gm_sum - tm.grad.cpu().double().numpy()).max() <= 1e-4 * np.abs(gm_sum).max()
    assert np.abs(gl_sum - tl.grad.cpu().double().numpy()).max() <= 1e-4 * np.abs(gl_sum).max()


def test_scene_api_gradient_descent(dev):
    """the reference's driver pattern through the class API: match_mirror.py
    loss, GDOptimizer, sphere centres move and the loss goes down."""
    import rrt
    from rrt import (Camera, GDOptimizer, Light, Material, PhongShader, Scene, Sphere, Square, rotate, shared,
                     translate)
    rrt.set_device(dev)
    c1 = shared(np.asarray([-.5, -.5, 4], dtype="float32"))
    c2 = shared(np.asarray([.5, .5, 4], dtype="float32"))
    m1 = Material((0.2, 0.9, 0.4), 0.3, 0.7, 0.5, 50.)
    m2 = Material((0.87, 0.1, 0.507), 0.3, 0.9, 0.4, 50.)

    def make_scene():
        objs = [Sphere(translate(c1), m1), Sphere(translate(c2), m2),
                Square(translate((0, 0, 3)) * rotate(50, [0., 1., 0.]), m2)]
        return Scene(objs, [Light((-1., -1., 2.), (1., 0.87, 0.961))], Camera(128, 128), PhongShader())

    render = make_scene().build(jitter=17).detach()
    flipped = torch.flip(render, dims=[1])                   # np.fliplr (match_mirror.py:40)
    cost = lambda: ((make_scene().build(jitter=17) - flipped) ** 2).sum()
    train = GDOptimizer().optimize([c1, c2], cost, 0.000008, 0.1)
    losses = [train() for _ in range(15)]
    assert losses[-1] < losses[0]
    assert not torch.equal(c1.detach().cpu(), torch.tensor([-.5, -.5, 4.]))


def test_c5_batch_256_orbit_scenes(dev):
    """BASELINE config 5 at its full size: 256 orbit scenes of 64x64, antialias_samples = 4, one batched
    launch.  B*S = 512 packed records exceed the kernels' LDS staging area, so they run scene-local: a
    workgroup takes whole list segments, stages the segment's scene and accumulates its gradient in LDS.
    Oracle on a sample of scenes; the shared material's gradient (d/dshininess included) summed over
    all 256 scenes."""
    import rrt
    from rrt import synthetic
    rs = np.random.RandomState(5)
    B, n = 256, 64
    _, mat, light, kinds = synthetic.orbit_scene()
    w = np.stack([synthetic.orbit_scene((rs.uniform(-9, 9), rs.uniform(-9, 9), 32.))[0] for _ in range(B)])
    cams = np.stack([synthetic.orbit_camera(2.5 if b % 2 == 0 else -2.5) for b in range(B)])
    G = rs.randn(B, n, n, 3).astype(np.float32)
    tw = torch.tensor(w, device=dev, requires_grad=True)
    tm = torch.tensor(mat, device=dev, requires_grad=True)
    tl = torch.tensor(light, device=dev, requires_grad=True)
    img = rrt.render(tw, tm, tl, kinds, torch.tensor(cams, device=dev), x_dims=n, y_dims=n, antialias_samples=4,
                     specular=False, jitter=4)
    (img * torch.tensor(G, device=dev)).sum().backward()
    torch.cuda.synchronize()
    gm = np.zeros((2, 7))
    gl = np.zeros(6)
    for b in range(B):
        oi, ow, om, ol = oracle(w[b], mat, light, kinds, G[b], x_dims=n, y_dims=n, antialias_samples=4, specular=False,
                                jitter=4, cam=cams[b])
        gm += om
        gl += ol
        if b in (0, 1, 37, 95, 128, 255):
            assert_image_close(img[b].detach().cpu().double().numpy(), oi, "image[%d]" % b)
            hwb = tw.grad[b].cpu().double().numpy()
            rel = np.abs(hwb - ow).max() / np.abs(ow).max()
            _note("g_w2o[%d]" % b, float(rel), 0, bool(rel > GRAD_TIGHT))
            if rel > GRAD_TIGHT:
                # (a grazing ray in this scene: the rule of the module docstring, on the scene rendered alone -- the
                # batched launch equals B single launches, test_batch_axis)
                assert rel <= GRAD_RTOL, "g_w2o[%d]: rel %.2e" % (b, rel)
                kw1 = dict(x_dims=n, y_dims=n, antialias_samples=4, specular=False, jitter=4, cam=cams[b])
                assert_grads_without_grazing_pixels(dev, w[b], mat, light, kinds, G[b].astype(np.float64), kw1, dict(kw1))
    # shared material / light gradients are summed over the batch
    assert_grad_close(tm.grad.cpu().double().numpy(), gm, "g_material summed over the batch", rtol=GRAD_TIGHT)
    # (the light's gradient is a sum over 256 scenes with opposite signs -- the two cameras look from either side:
    # float32 accumulation error relative to the cancelled total)
    assert_grad_close(tl.grad.cpu().double().numpy(), gl, "g_light summed over the batch", rtol=GRAD_RTOL)


def test_scene_local_mode_many_segments(dev):
    """scene-local mode with more list segments than workgroups and two segments per scene: 320 scenes of one
    shape, 64 columns x 300 rows (orbit camera: image rows run over y).  One batched launch == 320 launches
    (images bit for bit, per-scene gradients, the shared material's gradient summed)."""
    import rrt
    from rrt import synthetic
    rs = np.random.RandomState(8)
    B, nx, ny = 320, 64, 300
    base_w, mat, light, kinds = synthetic.orbit_scene()
    mat, kinds = mat[:1], kinds[:1]
    w = np.stack([synthetic.orbit_scene((rs.uniform(-6, 6), rs.uniform(-6, 6), 28.))[0][:1] for _ in range(B)])
    cam = synthetic.orbit_camera(2.5)
    G = rs.randn(B, ny, nx, 3).astype(np.float32)
    tw = torch.tensor(w, device=dev, requires_grad=True)
    tm = torch.tensor(mat, device=dev, requires_grad=True)
    tl = torch.tensor(light, device=dev, requires_grad=True)
    kw = dict(x_dims=nx, y_dims=ny, antialias_samples=1, specular=True, jitter=6)
    img = rrt.render(tw, tm, tl, kinds, torch.tensor(cam, device=dev)[None].expand(B, 16).contiguous(), **kw)
    assert img.shape == (B, ny, nx, 3)
    (img * torch.tensor(G, device=dev)).sum().backward()
    torch.cuda.synchronize()
    gm_sum = np.zeros((1, 7))
    for b in list(range(0, B, 23)) + [B - 1]:
        one = hip_render(dev, w[b], mat, light, kinds, grad=G[b], cam=cam, **kw)
        assert np.array_equal(one[0], img[b].detach().cpu().double().numpy()), b
        assert np.abs(one[1] - tw.grad[b].cpu().double().numpy()).max() <= 1e-4 * max(np.abs(one[1]).max(), 1e-20)
    for b in range(B):
        gm_sum += hip_render(dev, w[b], mat, light, kinds, grad=G[b], cam=cam, **kw)[2]
    assert np.abs(gm_sum - tm.grad.cpu().double().numpy()).max() <= 2e-4 * np.abs(gm_sum).max()
    assert (img.detach().amax(dim=(1, 2, 3)) > 0).float().mean() > 0.5       # the spheres are in view


def test_backward_twice_and_forward_only_reuse(dev):
    """retain_graph double backward gives the same gradients (accumulators are
    rewound by k_finalize); repeated forwards are bit-identical."""
    import rrt
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.random_spheres(6, seed=9)
    tw = torch.tensor(w2o, device=dev, requires_grad=True)
    tm = torch.tensor(mat, device=dev)
    tl = torch.tensor(light, device=dev)
    img = rrt.render(tw, tm, tl, kinds, x_dims=256, y_dims=256, antialias_samples=1)
    img2 = rrt.render(tw, tm, tl, kinds, x_dims=256, y_dims=256, antialias_samples=1)
    assert torch.equal(img, img2)
    loss = (img ** 2).sum()
    (g1,) = torch.autograd.grad(loss, tw, retain_graph=True)
    (g2,) = torch.autograd.grad(loss, tw)
    assert torch.allclose(g1, g2, rtol=1e-4, atol=1e-6 * g1.abs().max().item())


def test_example_planet_orbit_writes_reference_layout(dev, tmp_path):
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "orbit_dataset")
    subprocess.check_call([sys.executable, os.path.join(repo, "examples", "planet_orbit.py"), "-n", "6", "--out", out])
    d = np.load(out + ".npz")["arr_0"]
    assert d.shape == (6, 2, 64, 64, 3) and d.dtype == np.uint8
    # the fixed red sphere (0,0,48) x 6 is visible in every frame; the two cameras differ
    assert (d[..., 0] > 100).any() and not np.array_equal(d[:, 0], d[:, 1])


def test_graphed_gradient_descent_matches_eager(dev):
    """GDOptimizer(graph=True): the whole step captured in a HIP graph gives the
    same trajectory as the eager step (same kernels, same order)."""
    import rrt
    from rrt import Camera, GDOptimizer, Light, Material, PhongShader, Scene, Sphere, Square, rotate, shared, translate
    rrt.set_device(dev)

    def run(graph):
        c1 = shared(np.asarray([-.5, -.5, 4], dtype="float32"))
        c2 = shared(np.asarray([.5, .5, 4], dtype="float32"))
        m1 = Material((0.2, 0.9, 0.4), 0.3, 0.7, 0.5, 50.)
        m2 = Material((0.87, 0.1, 0.507), 0.3, 0.9, 0.4, 50.)
        light = Light((-1., -1., 2.), (1., 0.87, 0.961))
        cam = Camera(128, 128)

        def build():
            objs = [Sphere(translate(c1), m1), Sphere(translate(c2), m2),
                    Square(translate((0, 0, 3)) * rotate(50, [0., 1., 0.]), m2)]
            return Scene(objs, [light], cam, PhongShader()).build(jitter=17)
        flipped = torch.flip(build().detach(), dims=[1])
        cost = lambda: ((build() - flipped) ** 2).sum()
        train = GDOptimizer().optimize([c1, c2], cost, 0.000008, 0.1, graph=graph)
        losses = [train() for _ in range(8)]
        return losses, c1.detach().cpu().numpy(), c2.detach().cpu().numpy()

    le, c1e, c2e = run(False)
    lg, c1g, c2g = run(True)
    assert lg[-1] < lg[0]
    assert np.allclose(le, lg, rtol=1e-3) and np.allclose(c1e, c1g, atol=1e-4) and np.allclose(c2e, c2g, atol=1e-4)


def test_rectangular_image_and_orbit_band(dev):
    """x_dims != y_dims (the reference only works for squares: its jitter arrays
    are [x,y] added to [y,x] rays; the kernels and the oracle agree on the
    consistent reading) and a row band under the orbit camera (transpose = 0)."""
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.match_mirror(True)
    check_scene(dev, w2o, mat, light, kinds, x_dims=96, y_dims=160, antialias_samples=2, jitter=3)
    w2o, mat, light, kinds = synthetic.orbit_scene()
    cam = synthetic.orbit_camera(-2.5)
    check_scene(dev, w2o, mat, light, kinds, x_dims=64, y_dims=64, antialias_samples=2, specular=False, cam=cam,
                jitter=5, row0=20, rows=30)


def test_max_shapes_cull_vs_brute_force(dev):
    """S = RRT_MAX_SHAPES (1024 shapes, 16 mask words): culling changes no bit of
    the image, and the gradients agree with brute force up to atomic ordering."""
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.random_spheres(1024, seed=11)
    w2o[:, :, :3] *= 3.0                       # shrink the spheres 3x so the scene is not one solid wall
    w2o[:, :, 3] *= 3.0
    G = np.random.RandomState(2).randn(192, 192, 3)
    a = hip_render(dev, w2o, mat, light, kinds, grad=G, x_dims=192, y_dims=192, antialias_samples=1, cull=True)
    b = hip_render(dev, w2o, mat, light, kinds, grad=G, x_dims=192, y_dims=192, antialias_samples=1, cull=False)
    assert np.array_equal(a[0], b[0])
    for k in (1, 2, 3):
        assert np.abs(a[k] - b[k]).max() <= 1e-4 * max(np.abs(b[k]).max(), 1e-20)
    assert (a[0].max(-1) > 0).mean() > 0.05


@pytest.mark.parametrize("chunk", [1, 3])
def test_several_queue_batches(dev, chunk, monkeypatch):
    """the render kernels walk the list segments in batches (one batch at bench
    sizes); RRT_TUNE_CHUNK forces several at a test size.  Image bit-identical
    to the one-batch run, gradients equal up to float-atomic order."""
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.random_spheres(12, seed=3)
    G = np.random.RandomState(1).randn(640, 1024, 3)
    kw = dict(x_dims=640, y_dims=1024, antialias_samples=1)      # 16 x 3 list segments at P = 4
    ref = hip_render(dev, w2o, mat, light, kinds, grad=G, **kw)
    from rrt import _native
    monkeypatch.setenv("RRT_TUNE_CHUNK", str(chunk))
    _native.lib().rrt_debug_reload_tuning()           # the library reads RRT_TUNE_* once per process
    try:
        got = hip_render(dev, w2o, mat, light, kinds, grad=G, **kw)
    finally:
        monkeypatch.delenv("RRT_TUNE_CHUNK")
        _native.lib().rrt_debug_reload_tuning()
    assert np.array_equal(got[0], ref[0])
    for a, b, what in zip(got[1:], ref[1:], ("g_w2o", "g_material", "g_light")):
        assert_grad_close(a, b, what, rtol=1e-4)
    oi, ow, om, ol = oracle(w2o, mat, light, kinds, G, **kw)
    assert_image_close(got[0], oi, "image")
    assert_grad_close(got[1], ow, "g_w2o")


@pytest.mark.parametrize("shares,chunk", [("1:1", 0), ("3:1", 0), ("64:1", 0), ("7:5", 0), ("5:2", 3)])
def test_weighted_zero_tile_shares(dev, shares, chunk, monkeypatch):
    """k_fwd deals the tiles it only zero-fills with different shares for workgroups of even and odd blockIdx (the
    odd XCDs of an MI355X store ~15 % slower: DESIGN.md section 5; default 6:5).  Whatever the shares -- also
    across several queue batches -- every tile is written exactly once: the image equals the default's bit for
    bit, background included, and the gradients agree."""
    from rrt import synthetic, _native
    w2o, mat, light, kinds = synthetic.random_spheres(6, seed=4)
    w2o = w2o.copy()
    w2o[:, :, :3] *= 2.0                       # smaller spheres: most tiles have no candidate
    w2o[:, :, 3] *= 2.0
    G = np.random.RandomState(3).randn(1024, 2048, 3)
    kw = dict(x_dims=1024, y_dims=2048, antialias_samples=1)     # 8192 tiles: 256 workgroups, 32 list segments
    ref = hip_render(dev, w2o, mat, light, kinds, grad=G, **kw)
    assert (ref[0].max(-1) > 0).mean() < 0.5
    monkeypatch.setenv("RRT_TUNE_ZW", shares)
    if chunk:
        monkeypatch.setenv("RRT_TUNE_CHUNK", str(chunk))
    _native.lib().rrt_debug_reload_tuning()
    try:
        # poison the block the caching allocator hands to the render's output: an unwritten tile would read NaN
        poison = torch.full((1024, 2048, 3), float("nan"), device=dev)
        del poison
        got = hip_render(dev, w2o, mat, light, kinds, grad=G, **kw)
    finally:
        monkeypatch.delenv("RRT_TUNE_ZW")
        if chunk:
            monkeypatch.delenv("RRT_TUNE_CHUNK")
        _native.lib().rrt_debug_reload_tuning()
    assert np.array_equal(got[0], ref[0])
    for a, b, what in zip(got[1:], ref[1:], ("g_w2o", "g_material", "g_light")):
        assert_grad_close(a, b, what, rtol=1e-4)


@pytest.mark.parametrize("copies", [1, 2, 8])
def test_accumulator_copies(dev, copies, monkeypatch):
    """k_bwd's workgroups add their gradient sums to one of R copies of the accumulator rows (same-line float
    atomics serialise: DESIGN.md section 5; default R = 4) and k_finalize sums the copies and leaves them zeroed.
    Any R gives the same gradients -- one scene, and a batch with shared materials (summed over the scenes) -- also
    for a second backward on the same workspace (the copies must have been re-zeroed)."""
    import rrt
    from rrt import synthetic, _native
    w2o, mat, light, kinds = synthetic.random_spheres(9, seed=6)
    G = np.random.RandomState(4).randn(384, 512, 3)
    kw = dict(x_dims=384, y_dims=512, antialias_samples=2, jitter=5)

    def batch():
        tw = torch.tensor(np.stack([w2o, w2o[::-1].copy(), w2o]), device=dev, requires_grad=True)      # [3, S, 3, 4]
        tm = torch.tensor(mat, device=dev, requires_grad=True)                                         # shared
        tl = torch.tensor(light, device=dev, requires_grad=True)
        g = torch.tensor(np.random.RandomState(5).randn(3, 96, 96, 3).astype(np.float32), device=dev)
        out = []
        for _ in range(2):                         # two forward / backward pairs: fresh workspaces, then reused memory
            for t in (tw, tm, tl):
                t.grad = None
            img = rrt.render(tw, tm, tl, kinds, x_dims=96, y_dims=96, antialias_samples=1)
            (img * g).sum().backward()
            out.append([t.grad.cpu().double().numpy() for t in (tw, tm, tl)])
        torch.cuda.synchronize()
        return out
    ref, ref_b = hip_render(dev, w2o, mat, light, kinds, grad=G, **kw), batch()
    monkeypatch.setenv("RRT_TUNE_REPL", str(copies))
    _native.lib().rrt_debug_reload_tuning()
    try:
        got, got_b = hip_render(dev, w2o, mat, light, kinds, grad=G, **kw), batch()
    finally:
        monkeypatch.delenv("RRT_TUNE_REPL")
        _native.lib().rrt_debug_reload_tuning()
    assert np.array_equal(got[0], ref[0])
    for a, b, what in zip(got[1:], ref[1:], ("g_w2o", "g_material", "g_light")):
        assert_grad_close(a, b, what, rtol=1e-4)
    for rnd in range(2):
        for a, b, what in zip(got_b[rnd], ref_b[0], ("g_w2o", "g_material", "g_light")):
            assert_grad_close(a, b, "batch round %d %s" % (rnd, what), rtol=1e-4)


@pytest.mark.parametrize("samples,parts", [(4, 0), (4, 2), (3, 2), (3, 4), (2, 4), (5, 4)])
def test_sample_split(dev, samples, parts, monkeypatch):
    """on small images (tiles of 1 or 2 rows) k_bwd hands every queue entry out 2 or 4 times and each wave takes a share
    of the tile's antialiasing samples (KArgs::split_log2; default 4 shares from 4 samples on, else 2).  Any split --
    none, more shares than samples, samples that do not divide -- gives the same gradients: spheres + Square with hash
    jitter (multi- and single-candidate tiles, the generic adjoint), several queue batches, and a batch of small scenes
    whose workgroups take whole list segments"""
    import rrt
    from rrt import synthetic, _native
    w2o, mat, light, kinds = synthetic.match_mirror(True)
    G = np.random.RandomState(4).randn(200, 320, 3)
    kw = dict(x_dims=200, y_dims=320, antialias_samples=samples, jitter=9)
    ow, om, ol, kk = synthetic.orbit_scene()

    def batch():
        B = 160                                   # B * S > 128 records: lds_scene = 2
        ang = np.linspace(0.0, 2 * np.pi, B, endpoint=False)
        tw = torch.tensor(np.stack([synthetic.orbit_scene((9 * np.cos(t), 9 * np.sin(t), 32.0))[0] for t in ang]),
                          device=dev, requires_grad=True)
        tm = torch.tensor(om, device=dev, requires_grad=True)
        tl = torch.tensor(ol, device=dev, requires_grad=True)
        g = torch.tensor(np.random.RandomState(5).randn(B, 32, 32, 3).astype(np.float32), device=dev)
        cams = np.stack([synthetic.orbit_camera(2.5 if b % 2 == 0 else -2.5) for b in range(B)])
        img = rrt.render(tw, tm, tl, kk, torch.tensor(cams, device=dev), x_dims=32, y_dims=32, antialias_samples=samples,
                         specular=False, jitter=4)
        (img * g).sum().backward()
        torch.cuda.synchronize()
        return [t.grad.cpu().double().numpy() for t in (tw, tm, tl)]

    def run():
        return hip_render(dev, w2o, mat, light, kinds, grad=G, **kw), batch()
    monkeypatch.setenv("RRT_TUNE_SPLIT", "0")
    _native.lib().rrt_debug_reload_tuning()
    try:
        ref, ref_b = run()
        monkeypatch.setenv("RRT_TUNE_SPLIT", str(parts))
        if parts == 2:
            monkeypatch.setenv("RRT_TUNE_CHUNK", "3")
        _native.lib().rrt_debug_reload_tuning()
        got, got_b = run()
    finally:
        monkeypatch.delenv("RRT_TUNE_SPLIT")
        monkeypatch.delenv("RRT_TUNE_CHUNK", raising=False)
        _native.lib().rrt_debug_reload_tuning()
    assert np.array_equal(got[0], ref[0])
    for a, b, what in zip(got[1:], ref[1:], ("g_w2o", "g_material", "g_light")):
        assert_grad_close(a, b, what, rtol=1e-4)
    for a, b, what in zip(got_b, ref_b, ("g_w2o", "g_material", "g_light")):
        assert_grad_close(a, b, "batch " + what, rtol=1e-4)


def test_backward_with_and_without_saved_queues(dev, monkeypatch):
    """a backward on the forward's setup reads the queue batches the forward
    dealt; RRT_TUNE_NOSAVED makes it deal them again -- same gradients"""
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.random_spheres(20, seed=5)
    G = np.random.RandomState(2).randn(512, 512, 3)
    kw = dict(x_dims=512, y_dims=512, antialias_samples=2, jitter=11)
    a = hip_render(dev, w2o, mat, light, kinds, grad=G, **kw)
    from rrt import _native
    monkeypatch.setenv("RRT_TUNE_NOSAVED", "1")
    _native.lib().rrt_debug_reload_tuning()           # the library reads RRT_TUNE_* once per process
    try:
        b = hip_render(dev, w2o, mat, light, kinds, grad=G, **kw)
    finally:
        monkeypatch.delenv("RRT_TUNE_NOSAVED")
        _native.lib().rrt_debug_reload_tuning()
    assert np.array_equal(a[0], b[0])
    for x, y, what in zip(a[1:], b[1:], ("g_w2o", "g_material", "g_light")):
        assert_grad_close(x, y, what, rtol=1e-4)


def _random_scene(rs):
    """shapes with general affine world->object maps (rotation about a random axis, anisotropic
    scale, translation), a few of them Squares; random materials and light"""
    S = int(rs.randint(1, 7))
    w2o, mat, kinds = np.zeros((S, 3, 4)), np.zeros((S, 7)), np.zeros(S, np.uint8)
    for k in range(S):
        z = rs.uniform(3.0, 9.0)
        c = np.array([rs.uniform(-0.35, 0.35) * z, rs.uniform(-0.35, 0.35) * z, z])
        axis = rs.randn(3); axis /= np.linalg.norm(axis)
        ang = rs.uniform(0, np.pi)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        kinds[k] = 1 if rs.rand() < 0.25 else 0
        sc = rs.uniform(0.25, 0.9, size=3) * (z / 5.0) * (2.0 if kinds[k] else 1.0)
        o2w = R @ np.diag(sc)                                   # object -> world: translate * rotate * scale
        inv = np.linalg.inv(o2w)
        w2o[k, :, :3], w2o[k, :, 3] = inv, -inv @ c
        mat[k] = [*rs.uniform(0.05, 0.95, 3), rs.uniform(0.0, 0.5), rs.uniform(0.4, 0.9), rs.uniform(0.1, 0.5),
                  float(rs.choice([1.0, 2.0, 7.0, 20.0, 50.0]))]
    light = np.array([*rs.uniform(-1, 1, 2), rs.uniform(0.5, 2.0), *rs.uniform(0.5, 1.0, 3)])
    return w2o.astype(np.float32), mat.astype(np.float32), light.astype(np.float32), kinds


def _random_scene_case(dev, seed, centre):
    rs = np.random.RandomState(1000 + seed)
    w2o, mat, light, kinds = _random_scene(rs)
    x = int(rs.choice([33, 64, 100, 130, 257]))
    y = x if rs.rand() < 0.6 else int(rs.choice([48, 96, 192]))
    A = int(rs.randint(1, 4))
    shader = "depth" if seed % 4 == 3 else "phong"
    jit = int(rs.randint(1, 1 << 20))
    kw = dict(x_dims=x, y_dims=y, antialias_samples=A, jitter="center" if centre else jit, shader=shader,
              specular=bool(rs.rand() < 0.7), max_depth=12.0)
    if x != y or seed % 3 == 0:                               # the root camera of the reference only works for squares
        kw["cam"] = synthetic_orbit_cam(float(rs.uniform(-0.5, 0.5)))
    n_rows, n_cols = (y, x) if "cam" in kw else (x, y)
    G = rs.randn(n_rows, n_cols, 3)
    hi, hw, hm, hl = hip_render(dev, w2o, mat, light, kinds, grad=G, **kw)
    oi, ow, om, ol = oracle(w2o, mat, light, kinds, G, **kw)
    assert_image_close(hi, oi, "image")
    image_flips = int((np.abs(hi - oi).max(-1) > IMG_TOL).sum())
    if assert_grads_vs_oracle((hw, hm, hl), (ow, om, ol),
                              lambda: oracle(w2o, mat, light, kinds, G, dtype=torch.float32, **kw)[1:],
                              image_flips, lambda: oracle(w2o, mat, light, kinds, G, want_flips=True, **kw),
                              count_near=lambda: int((oracle(w2o, mat, light, kinds, None, want_margin=True, **kw) < GRAZING_MARGIN).sum())):
        assert_grads_without_grazing_pixels(dev, w2o, mat, light, kinds, G, kw, dict(kw))


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes_vs_oracle(dev, seed):
    """seeded sweep over what the boundary accepts: general affine shapes (Spheres and Squares),
    ragged image sizes on either side of the tile-row heuristic, 1..3 samples with hash jitter,
    Phong with / without specular and the depth shader, root and orbit camera.  Image against the
    float64 oracle; gradients against it up to the float32 noise floor the oracle itself shows
    when run in float32 at the same point (grazing rays)."""
    _random_scene_case(dev, seed, centre=False)


@pytest.mark.parametrize("seed", range(12, 20))
def test_random_scenes_centre_jitter(dev, seed):
    """the same sweep at the stratum centres: the centre-jitter kernels are instantiations of their own (JC = 1), and
    their backward runs squares, the depth shader and non-FAST materials through a called function (bwd_generic_shape)
    where the others have the code inline"""
    _random_scene_case(dev, seed, centre=True)


def synthetic_orbit_cam(y):
    from rrt import synthetic
    return synthetic.orbit_camera(y)
