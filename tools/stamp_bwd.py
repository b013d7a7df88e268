#!/usr/bin/env python3
"""Diagnostic: where does a k_bwd wave spend its cycles?  Builds a -DRRT_STAMPS
copy of the library (s_memtime stamps at phase boundaries, written to their own
buffer), runs the bench workload once and prints per-phase cycle statistics of
the non-empty tiles.  Read SHARES, not lengths (the stamps' fences forbid
overlaps the real kernel has).  Run on the GPU box:
    python tools/stamp_bwd.py [--spheres 32] [--size 4096]
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
                       "-DRRT_STAMPS", "-fno-slp-vectorize", "-I", os.path.join(REPO, "include"), "-I", os.path.join(PKG, "csrc"),
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
ap.add_argument("--onebig", action="store_true", help="one sphere covering 87 %% of the image: uniform single-candidate tiles")
args = ap.parse_args()
if args.onebig:
    from rrt import synthetic

    def one_big(S, seed=0):
        w2o = synthetic._w2o_translate_scale((0.0, 0.0, 4.0), 1.9)[None].astype(np.float32)
        return (w2o, np.array([[0.2, 0.9, 0.4, 0.3, 0.7, 0.5, 50.0]], np.float32),
                np.array([-1.0, -1.0, 2.0, 0.961, 1.0, 0.87], np.float32), np.zeros(1, np.uint8))
    synthetic.random_spheres = one_big
    args.spheres = 1
dev = torch.device("cuda:0")
job = bench.Job(dev, args.size, args.spheres, args.aa, 0, args.size, True)
job.fwd()
job.grad.copy_(2.0 * (job.image - torch.flip(job.image, dims=[1])))
for _ in range(3):
    job.fwd(); job.bwd()
torch.cuda.synchronize()
n_tiles = (args.size // 64) * (args.size // 4)
stamps = torch.zeros((n_tiles + 64, 16), dtype=torch.int64, device=dev)
job.fwd()
torch.cuda.synchronize()
rc = job.L.rrt_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))     # after the forward: k_fwd stamps the same buffer
assert rc == 0, rc
job.bwd()
torch.cuda.synchronize()
st = stamps.cpu().numpy()[:n_tiles]
heavy = st[:, 8] != 0
h = st[heavy].astype(np.float64)
print("tiles", n_tiles, "non-empty", heavy.sum())
# slots: 0 tile start | 1 gradient rows landed (vmcnt 0) | 2 next tile popped + its prefetch issued |
#        3 winners settled (multi-candidate tiles) | 4 shape loaded, accumulators zeroed | 5 rows done |
#        6 butterfly done | 7 LDS accumulation issued (last shape of the tile) | 8 tile end
#        13 workgroup id | 14 / 15 real time (100 MHz) at tile start / end
names = ["0->1 wait for this tile's gradient rows", "1->2 pop next tile + issue its prefetch", "2->3 winner search (multi) ",
         "3->4 shape load + accumulator init (last shape)", "4->5 adjoint rows (last shape)", "5->6 butterfly", "6->7 LDS accumulation",
         "7->8 tile epilogue"]
tot = h[:, 8] - h[:, 0]
print("total cycles/tile: median %.0f mean %.0f p90 %.0f" % (np.median(tot), tot.mean(), np.percentile(tot, 90)))
for i, nm in enumerate(names):
    a_, b_ = h[:, i], h[:, i + 1]
    ok = (a_ != 0) & (b_ != 0) & (b_ >= a_)
    d = (b_ - a_)[ok]
    if len(d):
        print("  %-50s median %8.0f mean %8.0f  share %5.1f%%  (n=%d)" % (nm, np.median(d), d.mean(),
                                                                         100 * d.sum() / tot.sum(), len(d)))
rt = (h[:, 15] - h[:, 14])            # s_memrealtime ticks (100 MHz) over the tile
ok = rt > 0
print("in-kernel clock: %.2f GHz (median over tiles of d(s_memtime)/d(s_memrealtime) x 100 MHz)" %
      (np.median((h[:, 8] - h[:, 0])[ok] / rt[ok]) * 0.1))
t_first, t_last = h[:, 14].min(), h[:, 15].max()
print("kernel span by s_memrealtime: %.1f us; sum of tile wave-times / span = %.2f waves per SIMD" %
      ((t_last - t_first) / 100.0, (rt[ok].sum() / (t_last - t_first)) / 1024.0))
single = h[:, 3] - h[:, 2] < 200
print("single-candidate tiles: %d; their adjoint rows median %.0f cycles, whole tile median %.0f" % (
    single.sum(), np.median((h[:, 5] - h[:, 4])[single]), np.median(tot[single])))
print("multi-candidate tiles: %d; winner search median %.0f, whole tile median %.0f" % (
    (~single).sum(), np.median((h[:, 3] - h[:, 2])[~single]), np.median(tot[~single])))
t0 = h[ok, 14] - t_first
t1 = h[ok, 15] - t_first
span = t_last - t_first
nb = 24
edges = np.linspace(0, span, nb + 1)
prof = []
for i in range(nb):
    lo, hi = edges[i], edges[i + 1]
    ov = np.clip(np.minimum(t1, hi) - np.maximum(t0, lo), 0, None).sum() / (hi - lo)
    prof.append(ov / 1024.0)
print("busy waves per SIMD over %d slices of the span: %s" % (nb, " ".join("%.1f" % x for x in prof)))
dur = (h[:, 15] - h[:, 14])[ok] / 100.0
print("tile duration (us): median %.2f p10 %.2f p90 %.2f p99 %.2f max %.2f" % (
    np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
ok = ok & (h[:, 13] < 65536)
wg = h[ok, 13].astype(np.int64)
dur = (h[:, 15] - h[:, 14])[ok] / 100.0
endt = (h[ok, 15] - t_first) / 100.0
last = np.zeros(wg.max() + 1)
np.maximum.at(last, wg, endt)
cntw = np.bincount(wg, minlength=len(last))
busy = np.zeros(len(last)); np.add.at(busy, wg, dur)
print("workgroup finish (us): min %.1f p10 %.1f median %.1f p90 %.1f max %.1f; tiles per workgroup %d..%d; "
      "sum of tile times per workgroup (us): min %.0f median %.0f max %.0f" % (
          last.min(), np.percentile(last, 10), np.median(last), np.percentile(last, 90), last.max(), cntw.min(), cntw.max(),
          busy.min(), np.median(busy), busy.max()))
print("tiles per workgroup: percentiles 0/10/50/90/100 = %s; workgroups seen %d" % (
    np.percentile(cntw, [0, 10, 50, 90, 100]).tolist(), (cntw > 0).sum()))
order = np.argsort(last)
print("the 8 last workgroups to finish: id %s tiles %s busy(us) %s" % (order[-8:].tolist(), cntw[order[-8:]].tolist(),
                                                                       np.round(busy[order[-8:]]).tolist()))
print("the 8 first: id %s tiles %s busy(us) %s" % (order[:8].tolist(), cntw[order[:8]].tolist(), np.round(busy[order[:8]]).tolist()))
print("correlation(finish time, tiles) = %.2f, (finish time, busy) = %.2f" % (np.corrcoef(last, cntw)[0, 1], np.corrcoef(last, busy)[0, 1]))
