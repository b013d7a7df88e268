"""the reference's two gradient-descent drivers end to end on the HIP path.

``match_mirror.py:16-53`` and ``optimize_brightness.py:19-58``, literally: the same scenes
through the class API (Scene / Sphere / Square / Material / Light / Camera / PhongShader /
translate / rotate / scale), 128 x 128, antialias_samples = 4, ONE jitter draw baked into every
build (the reference bakes its unseeded draw into the Theano graph; here a fixed hash seed so the
oracle can share it), GDOptimizer's update rule ``v <- v - lr * g`` (optimize.py:26-27: the
drivers pass a 4th "momentum" argument the reference ignores -- so does rrt.optimize), 90 steps.

Every 10 steps (and at the last one) the float64 oracle is evaluated at the GPU's current
parameters and must give the same image, loss and gradient.  The oracle is not free-running: a
grazing ray makes dLoss/dcentre jump by O(|g|) for a 1e-7 change of a centre, so two float
trajectories separate at the first such event; checking at the GPU's own parameters keeps the
comparison meaningful for all 90 steps.

Gradient tolerance: 1e-3 * max|g|; a checkpoint that misses it may pass on the
float32 noise floor of the formulas (4 x the error of the oracle run in float32 at the same point),
capped at 5e-3 * max|g| unless the winner of some ray differs between the float32 and float64 oracle
(``oracle.hit_ids``), i.e. unless a hit/miss flip at a silhouette is actually present.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import rrt_oracle as O

pytestmark = pytest.mark.gpu

N, A, SEED, STEPS = 128, 4, 1234, 90
TIGHT = 1e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from rrt import _native
    _native.lib()
    return torch.device("cuda:0")


def _oracle_checkpoint(w2o_of, centres, mat, light, kinds, loss_of, dtype):
    """image, loss, d loss / d centres of the oracle at `centres` (float32 values, evaluated in `dtype`)"""
    c = centres.to(dtype).clone().requires_grad_(True)
    sd = O.sample_dist_from_uniform(O.hash_u01(SEED, N, N, A))
    img = O.render(w2o_of(c, dtype), torch.tensor(mat, dtype=dtype), torch.tensor(light, dtype=dtype), list(kinds),
                   x_dims=N, y_dims=N, antialias_samples=A, sample_dist=sd, dtype=dtype)
    loss = loss_of(img)
    (g,) = torch.autograd.grad(loss, c)
    return img.detach().double(), float(loss), g.double()


def _flips(w2o_of, centres, kinds):
    """rays whose composite winner differs between the float32 and the float64 oracle"""
    sd = O.sample_dist_from_uniform(O.hash_u01(SEED, N, N, A))
    ids = [O.hit_ids(w2o_of(centres.to(dt), dt).detach(), list(kinds), x_dims=N, y_dims=N, antialias_samples=A,
                     sample_dist=sd, dtype=dt) for dt in (torch.float32, torch.float64)]
    return int((ids[0] != ids[1]).sum())


def _run_loop(dev, make_scene, centres, cost, lr, w2o_of, mat, light, kinds, loss_of, report):
    from rrt import GDOptimizer
    train = GDOptimizer().optimize(centres, cost, lr, 0.1)      # the drivers' call pattern (momentum ignored)
    losses = []
    for step in range(STEPS):
        if step % 10 == 0 or step == STEPS - 1:
            img = make_scene().build(A, jitter=SEED)
            loss = cost()
            g = torch.cat([x.reshape(1, 3) for x in torch.autograd.grad(loss, centres)])
            c_cpu = torch.stack([c.detach().cpu() for c in centres])
            img64, loss64, g64 = _oracle_checkpoint(w2o_of, c_cpu, mat, light, kinds, loss_of, torch.float64)
            err_img = (img.detach().cpu().double() - img64).abs().amax(-1)
            bad = int((err_img > 2e-4).sum())
            assert bad <= max(2, int(5e-4 * err_img.numel())), (step, bad, err_img.max().item())
            # pixel-sum losses: absolute tolerance from the image tolerance; SSE: relative
            assert abs(loss.item() - loss64) <= max(1e-5 * abs(loss64), 6 * 2e-4), (step, loss.item(), loss64)
            scale = g64.abs().max().item()
            err = (g.detach().cpu().double() - g64).abs().max().item()
            tol = TIGHT * scale
            flips = 0
            if err > tol:
                flips = _flips(w2o_of, c_cpu, kinds)
                _, _, g32 = _oracle_checkpoint(w2o_of, c_cpu, mat, light, kinds, loss_of, torch.float32)
                fl = 4.0 * (g32 - g64).abs().max().item()
                tol = max(tol, fl if flips else min(fl, 5e-3 * scale))
            report.append((step, err / max(scale, 1e-30), flips))
            assert err <= tol, "step %d: gradient error %.3g vs scale %.3g (%.2e rel), %d winner flips" % (
                step, err, scale, err / scale, flips)
        losses.append(train())
    return losses


def test_match_mirror_90_steps(dev):
    """match_mirror.py:16-53"""
    import rrt
    from rrt import Camera, Light, Material, PhongShader, Scene, Sphere, Square, rotate, shared, translate
    rrt.set_device(dev)
    try:
        center1 = shared(np.asarray([-.5, -.5, 4], dtype="float32"))
        center2 = shared(np.asarray([.5, .5, 4], dtype="float32"))
        material1 = Material((0.2, 0.9, 0.4), 0.3, 0.7, 0.5, 50.)
        material2 = Material((0.87, 0.1, 0.507), 0.3, 0.9, 0.4, 50.)
        light = Light((-1., -1., 2.), (1., 0.87, 0.961))
        camera = Camera(N, N)

        def make_scene():
            objs = [Sphere(translate(center1), material1), Sphere(translate(center2), material2),
                    Square(translate((0, 0, 3)) * rotate(50, [0., 1., 0.]), material2)]
            return Scene(objs, [light], camera, PhongShader())

        render = make_scene().build(A, jitter=SEED).detach()
        flipped = torch.flip(render, dims=[1])                       # np.fliplr (match_mirror.py:40)
        flipped64 = flipped.cpu().double()
        cost = lambda: ((make_scene().build(A, jitter=SEED) - flipped) ** 2).sum()

        sq = O.compose(O.translate((0., 0., 3.)), O.rotate(50., (0., 1., 0.)))

        def w2o_of(c, dtype):
            return torch.stack([O.translate(c[0], dtype)[1][:3], O.translate(c[1], dtype)[1][:3], sq[1][:3].to(dtype)])
        mat = np.array([[0.2, 0.9, 0.4, 0.3, 0.7, 0.5, 50.], [0.87, 0.1, 0.507, 0.3, 0.9, 0.4, 50.],
                        [0.87, 0.1, 0.507, 0.3, 0.9, 0.4, 50.]])
        lgt = np.array([-1., -1., 2., 1., 0.87, 0.961])
        report = []
        losses = _run_loop(dev, make_scene, [center1, center2], cost, 0.000008, w2o_of, mat, lgt, [0, 0, 1],
                           lambda im: ((im - flipped64.to(im.dtype)) ** 2).sum(), report)
        assert len(losses) == STEPS and losses[-1] < 0.9 * losses[0]
        moved = (torch.stack([center1, center2]).detach().cpu() -
                 torch.tensor([[-.5, -.5, 4.], [.5, .5, 4.]])).abs().max().item()
        assert moved > 0.05
        print("match_mirror 90 steps: loss %.4f -> %.4f; (step, rel grad err, flips) %s" % (losses[0], losses[-1], report))
    finally:
        rrt.config._device[0] = None


def test_optimize_brightness_90_steps(dev):
    """optimize_brightness.py:19-58 (its live pdb.set_trace() aside): gradient descent on minus the
    brightness of two probe pixels w.r.t. the centres of a sphere and a rotated, scaled ellipsoid"""
    import rrt
    from rrt import Camera, Light, Material, PhongShader, Scene, Sphere, rotate, scale, shared, translate
    rrt.set_device(dev)
    try:
        center1 = shared(np.asarray([-.5, -.5, 4], dtype="float32"))
        center2 = shared(np.asarray([.5, .5, 4], dtype="float32"))
        material1 = Material((0.2, 0.9, 0.4), 0.3, 0.7, 0.5, 50.)
        material2 = Material((0.87, 0.1, 0.507), 0.3, 0.9, 0.4, 50.)
        light = Light((-1., -1., 2.), (0.961, 1., 0.87))
        camera = Camera(N, N)

        def make_scene():
            t1 = translate(center1)
            t2 = translate(center2) * rotate(90, (0, 0, 1)) * scale((1, 2, 1.5))
            return Scene([Sphere(t1, material1), Sphere(t2, material2)], [light], camera, PhongShader())

        def cost():
            image = make_scene().build(A, jitter=SEED)
            return -image[90, 85].sum() - image[50, 90].sum()          # optimize_brightness.py:51

        def w2o_of(c, dtype):
            t2 = O.compose(O.compose(O.translate(c[1], dtype), O.rotate(90., (0., 0., 1.), dtype)),
                           O.scale((1., 2., 1.5), dtype))
            return torch.stack([O.translate(c[0], dtype)[1][:3], t2[1][:3]])
        mat = np.array([[0.2, 0.9, 0.4, 0.3, 0.7, 0.5, 50.], [0.87, 0.1, 0.507, 0.3, 0.9, 0.4, 50.]])
        lgt = np.array([-1., -1., 2., 0.961, 1., 0.87])
        report = []
        losses = _run_loop(dev, make_scene, [center1, center2], cost, 0.0008, w2o_of, mat, lgt, [0, 0],
                           lambda im: -im[90, 85].sum() - im[50, 90].sum(), report)
        assert len(losses) == STEPS and losses[-1] <= losses[0]
        print("optimize_brightness 90 steps: loss %.5f -> %.5f; (step, rel grad err, flips) %s" % (
            losses[0], losses[-1], report))
    finally:
        rrt.config._device[0] = None


@pytest.mark.parametrize("script,frames", [("match_mirror.py", ["0.png", "0lr.png", "1.png", "2.png", "3.png"]),
                                           ("optimize_brightness.py", ["0.png", "1.png", "2.png", "3.png"])])
def test_example_drivers_run(dev, tmp_path, script, frames):
    """the two example drivers (the reference's scripts on this package) execute and write their frames"""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "output")
    res = subprocess.run([sys.executable, os.path.join(repo, "examples", script), "--steps", "3", "--out", out],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    for f in frames:
        assert os.path.getsize(os.path.join(out, f)) > 0
    vals = [float(l) for l in res.stdout.splitlines() if l.strip() and l.strip().lstrip("-").replace(".", "", 1)
            .replace("e-", "", 1).replace("e+", "", 1).isdigit()]
    assert len(vals) == 3 and all(np.isfinite(vals))
