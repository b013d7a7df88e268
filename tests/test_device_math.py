"""The product's device math (reversible-raytracer_amd/csrc/rrt_device.h,
compiled for the host by tests/cpu_harness) against the oracle -- no GPU needed.

float64 instantiation: the hand-derived adjoint must agree with autograd over
the literal restatement to ~1e-9 (pins the derivation, the transposes, the
band logic, the hash jitter, Square, depth shader, orbit camera).
float32 instantiation: the noise floor the GPU tolerances are taken from.
"""
import numpy as np
import pytest
import torch

import harness_util as H
import rrt_oracle as O
from rrt import synthetic as syn


def _oracle(w2o, mat, light, kinds, G, n, A, jitter, cam, **kw):
    sd = None
    if jitter is not None:
        sd = O.sample_dist_from_uniform(O.hash_u01(jitter, n, n, A))
    cam_o2w = None
    if cam is not None:
        cam_o2w = np.eye(4)
        cam_o2w[:3, :4] = np.asarray(cam[:12]).reshape(3, 4)
    # shininess_grad="masked": the d/dshininess the product documents (include/rrt.h); nothing else changes
    return O.render_and_grads(w2o, mat, light, list(kinds), grad_image=G, x_dims=n, y_dims=n,
                              antialias_samples=A, sample_dist=sd, cam_o2w=cam_o2w, shininess_grad="masked", **kw)


CASES = {
    "match_mirror": dict(scene=syn.match_mirror, n=48, A=1),
    "match_mirror_aa_hash": dict(scene=syn.match_mirror, n=40, A=4, jitter=7),
    "match_mirror_depth": dict(scene=syn.match_mirror, n=40, A=2, shader="depth", max_depth=20.0),
    "random8": dict(scene=lambda: syn.random_spheres(8, seed=3), n=64, A=2, jitter=5),
    "random8_band": dict(scene=lambda: syn.random_spheres(8, seed=3), n=64, A=1, row0=17, rows=30),
    "orbit": dict(scene=syn.orbit_scene, n=48, A=4, jitter=11, cam=syn.orbit_camera(2.5), specular=False),
    "one_sphere": dict(scene=syn.one_sphere, n=32, A=1),
    # pure colours (a colour channel exactly 0, planet_orbit.py:26-33) lit from behind: phi < 0 on most of what the camera
    # sees -- the FAST_ZEROCH corner of the lean adjoint (the zero channel's d/dcolour = phi survives the clip at 0)
    "backlit_pure": dict(scene=lambda: _backlit_pure_scene(), n=40, A=2, jitter=4),
    "backlit_pure_nospec": dict(scene=lambda: _backlit_pure_scene(), n=40, A=2, jitter=4, specular=False),
}


def _backlit_pure_scene():
    w2o = np.stack([syn._w2o_translate_scale((-0.6, -0.2, 4.0), 1.0), syn._w2o_translate_scale((0.7, 0.4, 5.0), 1.2)])
    mat = np.array([[0.0, 0.9, 0.0, 0.3, 0.9, 0.1, 7.0], [0.8, 0.0, 0.6, 0.2, 0.7, 0.15, 20.0]])
    light = np.array([0.3, 0.2, -1.0, 1.0, 0.9, 0.8])
    return w2o.astype(np.float32), mat.astype(np.float32), light.astype(np.float32), np.zeros(2, dtype=np.uint8)


@pytest.mark.parametrize("name", sorted(CASES))
def test_adjoint_float64_matches_autograd(name):
    c = dict(CASES[name])
    w2o, mat, light, kinds = c.pop("scene")()
    n, A = c.pop("n"), c.pop("A")
    jitter, cam = c.pop("jitter", None), c.pop("cam", None)
    okw = {k: c[k] for k in ("shader", "specular", "max_depth", "row0", "rows") if k in c}
    rows = c.get("rows", n - c.get("row0", 0))
    G = np.random.RandomState(1).randn(rows, n, 3)
    oi, ow, om, ol = _oracle(w2o, mat, light, kinds, G, n, A, jitter, cam, **okw)
    hk = dict(x_dims=n, y_dims=n, A=A, cam=cam, transpose=cam is None, jitter=jitter, **okw)
    hi = H.fwd(w2o, mat, light, kinds, double=True, **hk)
    hw, hm, hl = H.bwd(w2o, mat, light, kinds, grad_image=G, double=True, **hk)
    assert np.abs(hi - oi.numpy()).max() < 1e-10

    def rel(a, b):
        b = b.numpy()
        return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert rel(hw, ow) < 1e-8 and rel(hl, ol) < 1e-8
    assert rel(hm[:, :6], om[:, :6]) < 1e-8 or np.abs(om[:, :6].numpy()).max() == 0
    # d/dshininess (column 6) against the masked-exponent oracle, per entry
    assert np.isfinite(om.numpy()).all()
    assert np.abs(hm[:, 6] - om[:, 6].numpy()).max() <= 1e-8 * max(np.abs(om[:, 6].numpy()).max(), 1e-30)


def _positive_base_scene():
    """one unit sphere so close that every pixel hits it near its pole, lit from the camera side:
    the specular base s = rm . look_at is > 0.7 on every ray, so ``s ** shininess`` is smooth in
    the shininess (no NaN lanes) and d/dshininess can be pinned by finite differences"""
    w2o = syn._w2o_translate_scale((0.02, -0.03, 1.25), 1.0)[None].astype(np.float32)
    mat = np.array([[.5, .6, .7, .4, .6, .3, 7.3]], np.float32)          # non-integer shininess
    light = np.array([0.1, -0.1, -1.0, .9, 1., .8], np.float32)
    return w2o, mat, light, np.zeros(1, np.uint8)


def test_shininess_gradient_against_finite_differences():
    """VERDICT r1 / ADVICE: d/dshininess was exported but compared with nothing.  On positive-base
    lanes Theano's own ``log(base) * pow`` is finite, so here: central finite differences of the
    float64 oracle's forward == oracle autograd (both readings) == the product's adjoint_ray."""
    w2o, mat, light, kinds = _positive_base_scene()
    n, A = 24, 2
    G = np.random.RandomState(0).randn(n, n, 3)
    sd = O.sample_dist_from_uniform(O.hash_u01(3, n, n, A))
    kw = dict(x_dims=n, y_dims=n, antialias_samples=A, sample_dist=sd)
    grads = {}
    for mode in ("theano", "masked"):
        img, gw, gm, gl = O.render_and_grads(w2o, mat, light, list(kinds), grad_image=G, shininess_grad=mode, **kw)
        assert np.isfinite(gm.numpy()).all()
        grads[mode] = gm.numpy()
    assert np.allclose(grads["theano"], grads["masked"], rtol=1e-12, atol=0)

    def f(m):
        im = O.render(w2o, m, torch.tensor(light, dtype=torch.float64), list(kinds), **kw)
        return (im * torch.tensor(G)).sum()
    fd = O.finite_difference_grad(f, torch.tensor(mat, dtype=torch.float64), eps=1e-6).numpy()
    assert abs(fd[0, 6]) > 1e-3                                       # the test has something to see
    assert np.abs(fd - grads["theano"]).max() <= 1e-6 * np.abs(fd).max()
    hw, hm, hl = H.bwd(w2o, mat, light, kinds, grad_image=G, x_dims=n, y_dims=n, A=A, jitter=3, double=True)
    assert np.abs(hm - grads["theano"]).max() <= 1e-8 * np.abs(fd).max()
    assert abs(hm[0, 6] - fd[0, 6]) <= 1e-6 * abs(fd[0, 6])
    # float32 instantiation: what the GPU can be held to
    hw32, hm32, hl32 = H.bwd(w2o, mat, light, kinds, grad_image=G, x_dims=n, y_dims=n, A=A, jitter=3, double=False)
    assert abs(hm32[0, 6] - fd[0, 6]) <= 1e-3 * abs(fd[0, 6])


@pytest.mark.parametrize("name", ["match_mirror_aa_hash", "random8", "orbit"])
def test_float32_noise_floor(name):
    """what float32 evaluation of the same formulas costs vs the float64 truth;
    the GPU tolerances (tests/test_gpu_parity.py) sit above these"""
    c = dict(CASES[name])
    w2o, mat, light, kinds = c.pop("scene")()
    n, A = c.pop("n"), c.pop("A")
    jitter, cam = c.pop("jitter", None), c.pop("cam", None)
    okw = {k: c[k] for k in ("shader", "specular", "max_depth", "row0", "rows") if k in c}
    G = np.random.RandomState(1).randn(n, n, 3)
    oi, ow, om, ol = _oracle(w2o, mat, light, kinds, G, n, A, jitter, cam, **okw)
    hk = dict(x_dims=n, y_dims=n, A=A, cam=cam, transpose=cam is None, jitter=jitter, **okw)
    hi = H.fwd(w2o, mat, light, kinds, double=False, **hk)
    hw, hm, hl = H.bwd(w2o, mat, light, kinds, grad_image=G, double=False, **hk)
    err = np.abs(hi - oi.numpy()).max(-1)
    assert (err > 2e-4).sum() <= 2
    assert np.abs(hw - ow.numpy()).max() <= 5e-3 * np.abs(ow.numpy()).max()
    assert np.abs(hl - ol.numpy()).max() <= 5e-3 * np.abs(ol.numpy()).max()


@pytest.mark.parametrize("S,n,A", [(32, 256, 1), (32, 128, 4), (150, 128, 1), (8, 64, 2)])
def test_tile_cull_is_conservative(S, n, A):
    """no ray of a tile may hit a shape the tile culled (64x4 and 64x1 tiles,
    centre and hash jitter, root and orbit cameras)"""
    w2o, mat, light, kinds = syn.random_spheres(S, seed=S)
    for th in (1, 4):
        for jit in (None, 3):
            v, mean = H.cull_check(w2o, mat, light, kinds, x_dims=n, y_dims=n, A=A, jitter=jit, tw=64, th=th)
            assert v == 0
            assert mean < S              # it does cull
    w2o, mat, light, kinds = syn.orbit_scene()
    v, _ = H.cull_check(w2o, mat, light, kinds, x_dims=64, y_dims=64, A=4, jitter=1, cam=syn.orbit_camera(-2.5),
                        transpose=False, tw=64, th=1)
    assert v == 0


def test_tile_cull_conservative_for_squares_and_ellipsoids():
    w2o, mat, light, kinds = syn.match_mirror(True)
    v, _ = H.cull_check(w2o, mat, light, kinds, x_dims=256, y_dims=256, A=2, jitter=9, tw=64, th=4)
    assert v == 0
    e = O.compose(O.compose(O.translate((.5, .5, 4.)), O.rotate(40., (0., 0., 1.))), O.scale((.2, .6, .06)))
    w = e[1].numpy()[None, :3].astype(np.float32)
    v, mean = H.cull_check(w, mat[:1], light, [0], x_dims=256, y_dims=256, A=1, tw=64, th=4)
    assert v == 0 and mean < 1


def test_degenerate_transform_is_never_culled():
    """a (near-)singular w2o cannot be bounded: the shape must stay a candidate"""
    w = np.zeros((1, 3, 4), np.float32)
    w[0, 0, 0] = 1.0
    w[0, 1, 1] = 1.0e-9
    w[0, 2, 2] = 1.0
    w[0, :, 3] = (0, 0, -4)
    mat = np.array([[.5, .5, .5, .3, .7, .5, 50.]], np.float32)
    light = np.array([0, 0, 1, 1, 1, 1], np.float32)
    v, mean = H.cull_check(w, mat, light, [0], x_dims=64, y_dims=64, A=1, tw=64, th=4)
    assert v == 0 and mean == 1.0


@pytest.mark.parametrize("name", ["match_mirror_aa_hash", "random8", "orbit", "one_sphere", "backlit_pure", "backlit_pure_nospec"])
def test_fast_paths_equal_generic_paths(name):
    """the lean code paths the kernels take for FAST_* shapes (shade_phong_fast, adjoint_sphere_phong:
    spheres with positive colour x intensity, shininess > 1) against the generic ones (shade_phong,
    adjoint_ray) in float64: same mathematics, so equal to rounding"""
    c = dict(CASES[name])
    w2o, mat, light, kinds = c.pop("scene")()
    n, A = c.pop("n"), c.pop("A")
    jitter, cam = c.pop("jitter", None), c.pop("cam", None)
    okw = {k: c[k] for k in ("specular",) if k in c}
    G = np.random.RandomState(5).randn(n, n, 3)
    hk = dict(x_dims=n, y_dims=n, A=A, cam=cam, transpose=cam is None, jitter=jitter, **okw)
    fi = H.fwd(w2o, mat, light, kinds, double=True, **hk)
    gi = H.fwd(w2o, mat, light, kinds, double=True, force_generic=True, **hk)
    assert np.abs(fi - gi).max() < 1e-12
    f = H.bwd(w2o, mat, light, kinds, grad_image=G, double=True, **hk)
    g = H.bwd(w2o, mat, light, kinds, grad_image=G, double=True, force_generic=True, **hk)
    for a, b in zip(f, g):
        assert np.abs(a - b).max() <= 1e-10 * max(np.abs(b).max(), 1e-30)


def test_fast_path_flags():
    """which shapes take the lean paths: Spheres with colour x intensity > 0; with the specular
    term only for shininess > 1.  Squares, dark or negative colours, small exponents stay generic --
    and the two must agree there too (the harness dispatches exactly like the kernels)"""
    w2o, mat, light, kinds = syn.match_mirror(True)
    n = 40
    G = np.random.RandomState(2).randn(n, n, 3)
    for sh, col in ((50.0, 0.5), (1.0, 0.5), (0.5, 0.5), (2.0, 0.0), (3.0, -0.2), (7.3, 0.5)):
        m = mat.copy()
        m[:, 6] = sh
        m[0, 0] = col
        oi, ow, om, ol = _oracle(w2o, m, light, kinds, G, n, 1, None, None)
        hi = H.fwd(w2o, m, light, kinds, x_dims=n, y_dims=n, A=1, double=True)
        ok = np.isfinite(oi.numpy())
        assert np.array_equal(ok, np.isfinite(hi))                     # pow() NaN lanes (non-integer exponent) agree
        assert np.abs(hi - oi.numpy())[ok].max() < 1e-10
        if ok.all():
            hw, hm, hl = H.bwd(w2o, m, light, kinds, grad_image=G, x_dims=n, y_dims=n, A=1, double=True)
            for a, b in ((hw, ow), (hm, om), (hl, ol)):
                assert np.abs(a - b.numpy()).max() <= 1e-8 * max(np.abs(b.numpy()).max(), 1e-30)


def _shadow_scene():
    """a big sphere with a small one between it and the light, a third one to the side, and a Square behind:
    the small sphere's shadow falls on the big one (about 10 % of its lit pixels), the Square receives
    shadows and casts none"""
    w2o = np.stack([syn._w2o_translate_scale((0.0, 0.0, 6.0), 1.5), syn._w2o_translate_scale((-0.9, -0.9, 4.2), 0.5),
                    syn._w2o_translate_scale((1.4, 0.3, 5.0), 0.4),
                    O.compose(O.translate((0.3, -0.2, 9.0)), O.scale((7.0, 7.0, 1.0)))[1].numpy()[:3]]).astype(np.float32)
    mat = np.array([[.8, .7, .6, .3, .7, .4, 20.], [.2, .8, .9, .2, .8, .3, 10.], [.9, .3, .3, .3, .6, .4, 50.],
                    [.5, .5, .8, .0, .6, .5, 1.]], np.float32)
    light = np.array([1., 1., 1.5, 1., .9, .8], np.float32)
    return w2o, mat, light, np.array([0, 0, 0, 1], np.uint8)


def test_shadows_extension_device_math_vs_oracle():
    """RRT_FLAG_SHADOWS (an extension: the reference's Sphere.shadow is dormant, shape.py:85-97, so nothing pins
    it): the product's shadowed_by + pack (object-space light direction) against the oracle's statement of the
    same reading, float64, image and all gradients; the shadow actually changes the picture"""
    w2o, mat, light, kinds = _shadow_scene()
    n, A = 64, 2
    G = np.random.RandomState(0).randn(n, n, 3)
    sd = O.sample_dist_from_uniform(O.hash_u01(5, n, n, A))
    kw = dict(x_dims=n, y_dims=n, antialias_samples=A, sample_dist=sd, shininess_grad="masked")
    plain = O.render(w2o, mat, light, list(kinds), **{k: v for k, v in kw.items() if k != "shininess_grad"})
    oi, ow, om, ol = O.render_and_grads(w2o, mat, light, list(kinds), grad_image=G, shadows=True, **kw)
    changed = (plain != oi).any(-1)
    assert 40 < int(changed.sum()) < 0.5 * n * n
    hi = H.fwd(w2o, mat, light, kinds, x_dims=n, y_dims=n, A=A, jitter=5, double=True, shadows=True)
    assert np.abs(hi - oi.numpy()).max() < 1e-10
    hw, hm, hl = H.bwd(w2o, mat, light, kinds, grad_image=G, x_dims=n, y_dims=n, A=A, jitter=5, double=True, shadows=True)
    for a, b in ((hw, ow), (hm, om), (hl, ol)):
        assert np.abs(a - b.numpy()).max() <= 1e-8 * max(np.abs(b.numpy()).max(), 1e-30)
    # shadows off: the flag changes nothing
    h0 = H.fwd(w2o, mat, light, kinds, x_dims=n, y_dims=n, A=A, jitter=5, double=True)
    assert np.abs(h0 - plain.numpy()).max() < 1e-10


def test_shadows_extension_gradient_finite_differences():
    """the shadow mask is piecewise constant, so away from mask flips the gradient of the shadowed render is the
    gradient of the lit pixels: central differences of the oracle's shadowed forward w.r.t. the material (a smooth
    dependence: no geometry moves, no pixel changes its mask) equal its autograd and the product's adjoint"""
    w2o, mat, light, kinds = _shadow_scene()
    n = 48
    G = np.random.RandomState(3).randn(n, n, 3)
    kw = dict(x_dims=n, y_dims=n, antialias_samples=1)
    oi, ow, om, ol = O.render_and_grads(w2o, mat, light, list(kinds), grad_image=G, shadows=True, shininess_grad="masked", **kw)

    def f(m):
        im = O.render(w2o, m, torch.tensor(light, dtype=torch.float64), list(kinds), shadows=True, **kw)
        return (im * torch.tensor(G)).sum()
    fd = O.finite_difference_grad(f, torch.tensor(mat, dtype=torch.float64), eps=1e-6).numpy()
    cols = [0, 1, 2, 3, 4, 5]                                          # (shininess: integer exponents, pow's sign rule changes off them)
    assert np.abs(fd[:, cols] - om.numpy()[:, cols]).max() <= 1e-6 * np.abs(fd[:, cols]).max()
    hw, hm, hl = H.bwd(w2o, mat, light, kinds, grad_image=G, x_dims=n, y_dims=n, A=1, double=True, shadows=True)
    assert np.abs(hm[:, cols] - fd[:, cols]).max() <= 1e-6 * np.abs(fd[:, cols]).max()
