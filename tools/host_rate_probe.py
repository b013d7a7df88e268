#!/usr/bin/env python3
"""How long does the HOST take to enqueue one bench step (two C-ABI calls, four launches)?  If that is well below the
GPU's time per step, the timed region of bench.py is GPU-bound from its second step on."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "reversible-raytracer_amd"))
import torch
import bench
dev = torch.device("cuda", 0)
job = bench.Job(dev, 4096, 32, 1, 0, 4096, True)
job.fwd(); job.grad.copy_(2.0 * (job.image - torch.flip(job.image, dims=[2])))
for _ in range(200):
    job.fwd(); job.bwd()
torch.cuda.synchronize()
for n in (1, 5, 20, 200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        job.fwd(); job.bwd()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("steps %3d: host enqueue %.1f us per step, until the GPU is done %.1f us per step" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
