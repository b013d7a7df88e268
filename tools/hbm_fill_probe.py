#!/usr/bin/env python3
"""How fast does this box write / read 201 MB (the bench image) with the simplest possible kernels?
Puts k_fwd's 12 B/pixel store stream and k_bwd's gradient read in context.  GPU box:
    python tools/hbm_fill_probe.py
"""
import torch

dev = torch.device("cuda:0")
n = 4096 * 4096 * 3
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
big = torch.empty(n * 8, dtype=torch.float32, device=dev)


def timeit(fn, reps=50, pre=None):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        if pre is not None:
            pre()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


mb = n * 4 / 1e6
for name, fn, pre in [
    ("fill 201 MB (same buffer every time)", lambda: x.zero_(), None),
    ("fill 201 MB after a 1.6 GB fill (caches flushed)", lambda: x.zero_(), lambda: big.zero_()),
    ("copy 201 MB -> 201 MB", lambda: y.copy_(x), None),
    ("read 201 MB (sum), just written", lambda: x.sum(), lambda: x.zero_()),
    ("read 201 MB (sum) after a 1.6 GB fill", lambda: x.sum(), lambda: big.zero_()),
]:
    med, best = timeit(fn, pre=pre)
    print("%-52s median %7.1f us  best %7.1f us  -> %.2f TB/s" % (name, med, best, mb / med))
