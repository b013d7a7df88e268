#!/usr/bin/env python3
"""Diagnostic: k_fwd wave timeline.  Builds a -DRRT_STAMPS copy of the library
(s_memrealtime at wave start, after the list prologue and at wave end, written
to their own buffer), runs the bench workload and prints when waves start, how
long the prologue takes, and how many waves are busy over the kernel's span.
Run on the GPU box:  python tools/stamp_fwd.py [--spheres 32] [--size 4096]
"""
import argparse
import ctypes
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "reversible-raytracer_amd")
OUT = os.path.join(REPO, "gpurun_out", "librrt_hip_stamps.so")
os.makedirs(os.path.dirname(OUT), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
                       "-DRRT_STAMPS", "-fno-slp-vectorize"] + os.environ.get("RRT_STAMP_FLAGS", "").split() + [ "-I", os.path.join(REPO, "include"), "-I", os.path.join(PKG, "csrc"),
                       os.environ.get("RRT_STAMP_SRC", os.path.join(PKG, "csrc", "rrt_kernels.hip")), "-o", OUT])
os.environ["RRT_LIB"] = OUT
sys.path.insert(0, PKG)
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--spheres", type=int, default=32)
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--aa", type=int, default=1)
args = ap.parse_args()
dev = torch.device("cuda:0")
job = bench.Job(dev, args.size, args.spheres, args.aa, 0, args.size, True)
for _ in range(3):
    job.fwd()
torch.cuda.synchronize()
n_rows = (args.size // 64) * args.size + 64
stamps = torch.zeros((n_rows, 8), dtype=torch.int64, device=dev)
assert job.L.rrt_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
job.fwd()
torch.cuda.synchronize()
st = stamps.cpu().numpy().astype(np.float64)
gw_all = np.arange(len(st))
gw_all = gw_all[st[:, 0] != 0]
st = st[st[:, 0] != 0]
t0 = st[:, 0].min()
start, pro, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0, (st[:, 2] - t0) / 100.0     # us (100 MHz)
span = end.max()
print("waves %d; span %.1f us" % (len(st), span))
print("wave start (us): median %.1f p90 %.1f max %.1f" % (np.median(start), np.percentile(start, 90), start.max()))
print("prologue (us): median %.2f p90 %.2f" % (np.median(pro - start), np.percentile(pro - start, 90)))
stg = (st[:, 5] - t0) / 100.0
print("  of which scene staging + first barrier (segment sizes in flight): median %.2f" % np.median(stg - start))
print("  segment-size scans + barrier: median %.2f; binary searches: median %.2f; entry gather + barrier: median %.2f" % (
    np.median((st[:, 6] - st[:, 5]) / 100.0), np.median((st[:, 7] - st[:, 6])[st[:, 7] != 0] / 100.0),
    np.median((st[:, 1] - st[:, 7])[st[:, 7] != 0] / 100.0)))
print("wave end (us): p10 %.1f median %.1f p90 %.1f max %.1f" % (np.percentile(end, 10), np.median(end),
                                                                 np.percentile(end, 90), end.max()))
work = end - pro
print("tile phase per wave (us): p10 %.1f median %.1f p90 %.1f max %.1f" % (np.percentile(work, 10), np.median(work),
                                                                            np.percentile(work, 90), work.max()))
print("tiles per wave: rendered mean %.2f max %d, zero-filled mean %.2f max %d" % (st[:, 3].mean(), st[:, 3].max(),
                                                                                  st[:, 4].mean(), st[:, 4].max()))
for h in range(int(st[:, 3].max()) + 1):
    sel = st[:, 3] == h
    if sel.sum():
        print("  waves with %d rendered tiles: %5d, tile phase median %.1f us" % (h, sel.sum(), np.median(work[sel])))
nb = 24
edges = np.linspace(0, span, nb + 1)
prof = []
for i in range(nb):
    lo, hi = edges[i], edges[i + 1]
    prof.append(np.clip(np.minimum(end, hi) - np.maximum(start, lo), 0, None).sum() / (hi - lo) / 1024.0)
print("resident waves per SIMD over %d slices of the span: %s" % (nb, " ".join("%.1f" % x for x in prof)))

# hardware XCD (HW_REG_XCC_ID, slot 3) against the block index
xcc = st[:, 3] - 1
print("XCC_ID of blocks 0..15:", [int(xcc[blk_i * 16]) if blk_i * 16 < len(xcc) else -1 for blk_i in range(16)])
for x in range(8):
    sel = xcc == x
    if sel.sum():
        print("XCC %d: %4d waves, blocks mod 8 = %s, wave end median %.1f p90 %.1f max %.1f us" % (
            x, sel.sum(), sorted(set(((gw_all[sel] // 16) % 8).tolist())), np.median(end[sel]), np.percentile(end[sel], 90), end[sel].max()))
# by XCD (workgroups are dealt round-robin over the 8 XCDs: blocks b and b + 8 share one) and by workgroup
blk = gw_all // 16
for x in range(8):
    sel = blk % 8 == x
    print("blocks = %d mod 8: wave end median %.1f p90 %.1f max %.1f us" % (x, np.median(end[sel]), np.percentile(end[sel], 90), end[sel].max()))
wg_end = np.array([end[blk == b].max() for b in np.unique(blk)])
wg_first = np.array([end[blk == b].min() for b in np.unique(blk)])
print("workgroup end (last wave): p10 %.1f median %.1f p90 %.1f max %.1f; first wave out of a workgroup: median %.1f" % (
    np.percentile(wg_end, 10), np.median(wg_end), np.percentile(wg_end, 90), wg_end.max(), np.median(wg_first)))
