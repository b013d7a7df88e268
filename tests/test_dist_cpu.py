"""Multi-process (world_size 2, gloo, CPU) test of the row-band sharding logic
of rrt.dist: band partition, one all-reduce of the flat gradient buffer, the
image gather.  The render itself is HIP-only, so here -- in the tests only --
the CPU oracle stands in for the band render; everything else is the product
code path (rrt.dist.band / all_reduce_flat / gather_image, RenderFn's hook)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(repo, "reversible-raytracer_amd"), os.path.join(repo, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rrt_oracle as O
    from rrt import dist as rdist
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.match_mirror(True)
    n = 37                                             # not divisible by 2: uneven bands
    row0, rows = rdist.band(n, rank, world)
    G = np.random.RandomState(0).randn(n, n, 3)
    img, gw, gm, gl = O.render_and_grads(w2o, mat, light, list(kinds), grad_image=G[row0:row0 + rows],
                                         x_dims=n, y_dims=n, antialias_samples=2, row0=row0, rows=rows)
    flat = torch.cat([gw.reshape(-1), gm.reshape(-1), gl.reshape(-1)]).float()
    # the ONE collective of a sharded backward, through the hook RenderFn.backward calls on its flat
    # [g_w2o | g_material | g_light] buffer (rrt/render.py) -- installed by the product's context manager
    import importlib
    render_mod = importlib.import_module("rrt.render")
    assert render_mod._grad_hook[0] is None
    with rdist.sharded():
        assert render_mod._grad_hook[0](flat) is flat
    assert render_mod._grad_hook[0] is None
    full = rdist.gather_image(img.float(), n)
    np.save(os.path.join(out_dir, "flat_%d.npy" % rank), flat.numpy())
    np.save(os.path.join(out_dir, "img_%d.npy" % rank), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_band_partition_covers_rows_exactly():
    from rrt.dist import band
    for n in (1, 7, 64, 4096, 4097):
        for ws in (1, 2, 3, 8):
            spans = [band(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and sum(s[1] for s in spans) == n
            for a, b in zip(spans, spans[1:]):
                assert a[0] + a[1] == b[0]
            assert max(s[1] for s in spans) - min(s[1] for s in spans) <= 1


def test_two_rank_sharding_reproduces_single_process(tmp_path):
    import rrt_oracle as O
    from rrt import synthetic
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w2o, mat, light, kinds = synthetic.match_mirror(True)
    n = 37
    G = np.random.RandomState(0).randn(n, n, 3)
    img, gw, gm, gl = O.render_and_grads(w2o, mat, light, list(kinds), grad_image=G, x_dims=n, y_dims=n,
                                         antialias_samples=2)
    ref = torch.cat([gw.reshape(-1), gm.reshape(-1), gl.reshape(-1)]).numpy()
    f0, f1 = np.load(tmp_path / "flat_0.npy"), np.load(tmp_path / "flat_1.npy")
    assert np.array_equal(f0, f1, equal_nan=True)      # every rank holds the same summed gradient (the oracle's d/dshininess is NaN, as Theano's)
    finite = np.isfinite(ref)
    assert np.allclose(f0[finite], ref[finite], rtol=1e-4, atol=1e-4 * np.abs(ref[finite]).max())
    i0, i1 = np.load(tmp_path / "img_0.npy"), np.load(tmp_path / "img_1.npy")
    assert np.array_equal(i0, i1) and i0.shape == (n, n, 3)
    assert np.allclose(i0, img.numpy(), atol=1e-6)


def test_sharded_context_hooks_the_flat_gradient():
    """rrt.dist.sharded() installs the all-reduce on RenderFn's flat buffer and
    restores the previous hook on exit (single process: the hook is a no-op sum)."""
    import importlib
    from rrt import dist as rdist
    render = importlib.import_module("rrt.render")
    assert render._grad_hook[0] is None
    with rdist.sharded():
        hook = render._grad_hook[0]
        assert hook is not None
        t = torch.ones(5)
        assert hook(t) is t and torch.equal(t, torch.ones(5))
    assert render._grad_hook[0] is None
