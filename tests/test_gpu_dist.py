"""RCCL on the device: the row-band sharding of rrt.dist with the "nccl" (= RCCL) backend.

A one-GPU box can only form a world of size 1 (RCCL refuses two ranks on one device), so this
checks that the product path -- rrt.dist.sharded() + render_band + RenderFn.backward's hook + a
real RCCL all_reduce on the flat gradient buffer, on the render stream -- runs on the GPU and
leaves the single-process result unchanged.  world_size 2 is covered on CPU with gloo
(tests/test_dist_cpu.py); N = 2, 4, 8 GPUs is what ``bench.py --gpus N`` runs on a full node."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    yield dist
    dist.destroy_process_group()


def test_sharded_render_band_backward_over_rccl(nccl_world1):
    dist = nccl_world1
    import rrt
    from rrt import Camera, Light, Material, PhongShader, Scene, Sphere, Square, rotate, shared, translate
    from rrt import dist as rdist
    assert dist.get_backend() == "nccl"
    dev = torch.device("cuda:0")
    rrt.set_device(dev)
    try:
        c1 = shared(np.asarray([-.5, -.5, 4], dtype="float32"))
        c2 = shared(np.asarray([.5, .5, 4], dtype="float32"))
        m1 = Material((0.2, 0.9, 0.4), 0.3, 0.7, 0.5, 50.)
        m2 = Material((0.87, 0.1, 0.507), 0.3, 0.9, 0.4, 50.)

        def scene():
            objs = [Sphere(translate(c1), m1), Sphere(translate(c2), m2),
                    Square(translate((0, 0, 3)) * rotate(50, [0., 1., 0.]), m2)]
            return Scene(objs, [Light((-1., -1., 2.), (1., 0.87, 0.961))], Camera(96, 96), PhongShader())

        target = torch.flip(scene().build(2, jitter=7).detach(), dims=[1])
        # single process, no collective
        loss = ((scene().build(2, jitter=7) - target) ** 2).sum()
        g_ref = torch.autograd.grad(loss, [c1, c2])
        # the sharded path: this rank's band (all rows at world size 1), gradient all-reduced over RCCL
        calls = []
        real = dist.all_reduce

        def counting(t, *a, **k):
            calls.append(t.numel())
            return real(t, *a, **k)
        dist.all_reduce = counting
        try:
            with rdist.sharded():
                band, row0, rows = rdist.render_band(scene(), 2, jitter=7)
                assert (row0, rows) == (0, 96) and band.shape == (96, 96, 3)
                loss_b = ((band - target[row0:row0 + rows]) ** 2).sum()
                # RenderFn.backward hands its flat [g_w2o | g_material | g_light] buffer to the hook:
                # 3 shapes x 12 + 3 x 7 + 6 floats through ONE RCCL all_reduce
                part = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss_b, [c1, c2])])
                lt = loss_b.detach().reshape(1).clone()
                dist.all_reduce(lt)                      # the loss partial, as a sharded driver sums it
        finally:
            dist.all_reduce = real
        torch.cuda.synchronize()
        assert calls == [3 * 12 + 3 * 7 + 6, 1]
        assert torch.allclose(part, torch.cat([g.reshape(-1) for g in g_ref]), rtol=1e-4, atol=1e-7)
        assert torch.allclose(lt[0], loss.detach(), rtol=1e-6)
        full = rdist.gather_image(band.detach(), 96)
        assert torch.equal(full, band.detach())
    finally:
        rrt.config._device[0] = None


def test_flat_gradient_buffer_all_reduce_on_render_stream(nccl_world1):
    """the flat [g_w2o | g_material | g_light] buffer the backward kernels fill is what goes into
    RCCL, straight from the stream the kernels ran on (no host round trip)"""
    dist = nccl_world1
    import rrt
    from rrt import render as _r   # noqa: F401  (rrt.render the function shadows the module)
    import importlib
    render_mod = importlib.import_module("rrt.render")
    from rrt import synthetic
    dev = torch.device("cuda:0")
    w2o, mat, light, kinds = synthetic.random_spheres(8, seed=1)
    tw = torch.tensor(w2o, device=dev, requires_grad=True)
    tm = torch.tensor(mat, device=dev, requires_grad=True)
    tl = torch.tensor(light, device=dev, requires_grad=True)
    G = torch.tensor(np.random.RandomState(0).randn(128, 128, 3).astype(np.float32), device=dev)
    img = rrt.render(tw, tm, tl, kinds, x_dims=128, y_dims=128, antialias_samples=1)
    ref = torch.autograd.grad((img * G).sum(), [tw, tm, tl], retain_graph=True)
    seen = []
    prev = render_mod._grad_hook[0]
    render_mod._grad_hook[0] = lambda flat: (seen.append(flat.numel()), dist.all_reduce(flat))
    try:
        got = torch.autograd.grad((img * G).sum(), [tw, tm, tl])
    finally:
        render_mod._grad_hook[0] = prev
    torch.cuda.synchronize()
    assert seen == [8 * 12 + 8 * 7 + 6]
    for a, b in zip(got, ref):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 * b.abs().max().item())
