#!/usr/bin/env python3
"""Expected strong-scaling curve of the 4096x4096 / 32-sphere workload from ONE GPU: the row band of every
rank of an N-GPU run (N = 1, 2, 4, 8) is rendered on this GPU, one after the other, and timed -- a step is
as slow as its slowest rank, plus the all-reduce of the 646-float gradient buffer (measured separately with
RCCL at world size 1: launch + kernel latency; the 8-GPU figure over xGMI is the driver's to measure).
    python tools/band_cost_model.py [--size 4096] [--spheres 32] [--steps 200]
Prints one JSON line per N: per-rank step times (us), the slowest, the fixed cost, the modelled speed-up."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "reversible-raytracer_amd"))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from rrt.dist import band, balanced_bands, rebalance_bands, row_cost_from_coverage  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--spheres", type=int, default=32)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--balanced", action="store_true", help="bands of equal estimated cost (bench.py's default for N > 1) instead of equal rows")
args = ap.parse_args()
args.no_cull, args.workload, args.aa = False, "c4", 1
dev = torch.device("cuda:0")


def time_band(row0, rows):
    job = bench.Job(dev, args.size, args.spheres, 1, row0, rows, True)
    job.fwd()
    job.grad.copy_(2.0 * (job.image - torch.flip(job.image, dims=[2])))
    for _ in range(20):
        job.fwd(); job.bwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.fwd(); job.bwd()
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / args.steps * 1e6
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    ks = []
    for _ in range(30):
        ev[0].record(); job.fwd(); ev[1].record(); job.bwd(); ev[2].record()
        torch.cuda.synchronize()
        ks.append((ev[0].elapsed_time(ev[1]) * 1e3, ev[1].elapsed_time(ev[2]) * 1e3))
    ks = np.median(np.array(ks), axis=0)
    return step, float(ks[0]), float(ks[1])


base = None
cost = row_cost_from_coverage(bench.probe_coverage(dev, args), args.size) if args.balanced else None
for n in (1, 2, 4, 8):
    per_rank = []
    bands = balanced_bands(cost, n) if args.balanced else [band(args.size, r, n) for r in range(n)]
    if args.balanced and n > 1:
        # bench.py's procedure (balanced_bands_of), with this one GPU standing in for every rank in turn
        best = None
        for _ in range(3):
            times = [bench.time_band(dev, args, args.size, *bands[r]) for r in range(n)]
            if best is None or max(times) < best[0]:
                best = (max(times), bands)
            bands = rebalance_bands(bands, times)
            if bands == best[1]:
                break
        bands = best[1]
    for r in range(n):
        row0, rows = bands[r]
        step, tf, tb = time_band(row0, rows)
        per_rank.append({"rank": r, "rows": rows, "step_us": round(step, 1), "fwd_call_us": round(tf, 1), "bwd_call_us": round(tb, 1)})
    slow = max(p["step_us"] for p in per_rank)
    if base is None:
        base = slow
    print(json.dumps({"n_gpus": n, "slowest_rank_us": slow, "mean_rank_us": round(float(np.mean([p["step_us"] for p in per_rank])), 1),
                      "speedup_without_allreduce": round(base / slow, 2), "ranks": per_rank}))
