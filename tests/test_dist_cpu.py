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


def test_balanced_bands_partition():
    """bands of equal estimated cost (strong scaling, bench.py's default for N > 1): contiguous, cover every row once,
    boundaries on the tile height, nobody without rows while rows last, equal cost where the rows allow it; degenerate
    inputs fall back to equal row counts"""
    from rrt.dist import balanced_bands, band, row_cost_from_coverage
    rs = np.random.RandomState(0)
    for n, world in ((4096, 8), (4096, 3), (1000, 4), (130, 6), (64, 8)):
        cov = np.clip(rs.rand(97) * (rs.rand(97) < 0.4), 0, 1)            # a probe of another height than the image
        cost = row_cost_from_coverage(cov, n)
        assert cost.shape == (n,) and (cost >= 1.0).all()
        bands = balanced_bands(cost, world)
        assert len(bands) == world and bands[0][0] == 0 and sum(b for _, b in bands) == n
        for r in range(world - 1):
            assert bands[r][0] + bands[r][1] == bands[r + 1][0] and bands[r + 1][0] % 4 == 0
        assert all(b >= 1 for _, b in bands)
        share = np.array([cost[a:a + b].sum() for a, b in bands]) / cost.sum() * world
        if n >= 1000:
            assert np.abs(share - 1.0).max() < 0.05
    # a heavy lower half: the cut moves up
    (a0, r0), (a1, r1) = balanced_bands(np.r_[np.ones(512), 3 * np.ones(512)], 2)
    assert (a0, r0 + r1) == (0, 1024) and r0 > 640
    # degenerate: fewer rows than ranks x tile height, zero or non-finite costs, one rank, no rows
    assert balanced_bands(np.ones(7), 8) == [band(7, r, 8) for r in range(8)]
    assert balanced_bands(np.zeros(64), 4) == [band(64, r, 4) for r in range(4)]
    assert balanced_bands(np.full(64, np.nan), 4) == [band(64, r, 4) for r in range(4)]
    assert balanced_bands(np.ones(64), 1) == [(0, 64)]
    assert balanced_bands([], 3) == [(0, 0)] * 3


def test_rebalance_bands_moves_rows_off_the_slow_rank():
    """measured-cost balancing (bench.py --gpus N, before the timed region): a rank slower than the others gives rows
    away, the partition stays exact, equal times change nothing much, and iterating on a synthetic cost converges"""
    from rrt.dist import band, rebalance_bands
    bands = [band(4096, r, 4) for r in range(4)]
    new = rebalance_bands(bands, [33.9, 40.1, 33.7, 33.4])
    assert sum(b for _, b in new) == 4096 and new[0][0] == 0
    for r in range(3):
        assert new[r][0] + new[r][1] == new[r + 1][0] and new[r + 1][0] % 4 == 0
    assert new[1][1] < 1024 and all(new[r][1] > 1024 for r in (0, 2, 3))
    same = rebalance_bands(bands, [35.0, 35.0, 35.0, 35.0])
    assert max(abs(b - 1024) for _, b in same) <= 4
    # a hidden per-row cost (a heavy stripe) + a fixed cost per rank: three rounds bring the slowest rank near the mean
    true = np.ones(4096)
    true[1200:1900] = 4.0

    def measure(bs):
        return [24.0 + 50.0 * true[a:a + b].sum() / true.sum() for a, b in bs]
    t = measure(bands)
    first = max(t) - min(t)
    for _ in range(3):
        bands = rebalance_bands(bands, t)
        t = measure(bands)
    assert max(t) - min(t) < 0.2 * first


def _balance_worker(rank, world, port, out_dir):
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (repo, os.path.join(repo, "reversible-raytracer_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    true = np.ones(1024)
    true[300:520] = 5.0                                # rows that cost five times the others (and show no extra coverage)

    def probe(dev, args):
        # rank 1 sees a last-bit-different probe: the first cut must still be rank 0's on every rank
        return np.full(64, 0.25) + (1e-7 if rank == 1 else 0.0) * np.arange(64)

    def timer(dev, args, full_rows, row0, rows):
        return 24.0 + 60.0 * true[row0:row0 + rows].sum() / true.sum()
    bands, times = bench.balanced_bands_of(torch.device("cpu"), None, 1024, rank, world, dist, probe=probe, timer=timer)
    np.save(os.path.join(out_dir, "bands_%d.npy" % rank), np.array(bands))
    np.save(os.path.join(out_dir, "times_%d.npy" % rank), np.array(times))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_band_balancing_choreography(tmp_path):
    """bench.py's search for bands of equal measured cost, over gloo at world size 3 with a synthetic probe and timer:
    every rank ends with the same exact partition (rank 0's first cut is broadcast, later cuts come from all-gathered
    times), the collectives line up (no hang), and the slowest rank is faster than with equal row counts"""
    port = _free_port()
    mp.spawn(_balance_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    b = [np.load(tmp_path / ("bands_%d.npy" % r)) for r in range(3)]
    t = [np.load(tmp_path / ("times_%d.npy" % r)) for r in range(3)]
    assert np.array_equal(b[0], b[1]) and np.array_equal(b[0], b[2]) and np.array_equal(t[0], t[1]) and np.array_equal(t[0], t[2])
    assert b[0][0, 0] == 0 and b[0][:, 1].sum() == 1024 and (b[0][:-1, 0] + b[0][:-1, 1] == b[0][1:, 0]).all()
    true = np.ones(1024)
    true[300:520] = 5.0
    equal = max(24.0 + 60.0 * true[r * 1024 // 3:(r + 1) * 1024 // 3].sum() / true.sum() for r in range(3))
    assert t[0].max() < equal - 5.0


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
