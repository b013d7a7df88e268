#!/usr/bin/env python3
"""Where a step's time goes, by call pattern and band height (GPU box).

    RRT_LIB=build/variants/librrt_X.so python tools/mall_probe.py [--rows 4096,512]

For every band height: the render kernels alone back to back (the image stays in the
Infinity Cache when it fits), alternating as in a step, and the whole calls (with k_prep /
k_finalize), all by HIP events on the launch stream.
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "reversible-raytracer_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def timed(fn, n=100):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3


def loop(fn, n=200):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="4096,2048,1024,512,256")
    ap.add_argument("--size", type=int, default=4096)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for rows in [int(x) for x in args.rows.split(",")]:
        row0 = (args.size - rows) // 2
        job = bench.Job(dev, args.size, 32, 1, row0, rows, True)
        job.fwd()
        job.grad.copy_(2.0 * (job.image - torch.flip(job.image, dims=[2])))
        for _ in range(300):
            job.fwd()
            job.bwd()
        out = {"lib": os.environ.get("RRT_LIB", "default"), "rows": rows}
        out["step_loop_us"] = loop(lambda: (job.fwd(), job.bwd()))
        out["k_fwd_alone_loop_us"] = loop(lambda: job.fwd(reuse=True))
        out["k_bwd_alone_loop_us"] = loop(lambda: job.bwd(kernel_only=True))
        job.fwd()
        out["alt_kernels_loop_us"] = loop(lambda: (job.fwd(reuse=True), job.bwd(kernel_only=True)))
        job.fwd()
        out["fwd_call_loop_us"] = loop(lambda: job.fwd())
        out["bwd_call_loop_us"] = loop(lambda: job.bwd())
        t_f, t_b = bench.time_kernels(job, 100)
        out["k_fwd_in_alt_us"] = t_f * 1e6
        out["k_bwd_in_alt_us"] = t_b * 1e6
        print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items()}), flush=True)
        del job
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
