#!/usr/bin/env python3
"""bench.py -- Mrays/s of the render hot path (forward + analytic backward).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 4096] [--spheres 32] [--aa 1]

A "step" is one pass of the hot path over one batch of synthetic input: one
``rrt_render_fwd`` + one ``rrt_render_bwd`` through the C ABI (include/rrt.h)
with a dense upstream gradient, inputs resident in HBM; for N > 1 the image's
row bands are sharded over the ranks (one process per GPU, launched by
``python -m torch.distributed.run``) and every step ends with ONE all-reduce of
the flat parameter-gradient buffer over RCCL (strong scaling of the fixed
4096x4096 image, BASELINE.json configs[3]).

Workload: seed 0, 32 random spheres translate(c)*scale(rho),
Phong with specular, root camera, 4096x4096, antialias_samples=1 (jitter at the
pixel centre), upstream grad g = 2*(image - fliplr(image0)) computed once
(match_mirror.py:40,45 loss).

Prints ONE JSON line (rank 0).  ``roofline`` is for the dominant kernel,
timed live with HIP events on the launch stream; ``cpu_baseline`` is the CPU
oracle (float32 torch, array-at-a-time like the Theano graph) timed on this
host on a bounded band of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "reversible-raytracer_amd"))
# Kernel arguments in device memory (the HIP runtime's default on this image; stated here so that the figure does not depend on
# it): with host-resident kernargs every workgroup's first scalar loads cross PCIe -- +2.3 us per kernel of the chain,
# 70.6 -> 79.7 us per step.  Must be in the environment before the runtime initialises.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
BYTES_PER_PIXEL_FWD = 12       # RGB float32 image write
BYTES_PER_PIXEL_BWD = 12       # RGB float32 upstream-gradient read


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup-ms", type=float, default=50.0,
                    help="untimed device spin-up before the W warmup steps (the clocks settle over the first tens of "
                         "milliseconds of work; reported in config.spinup_ms)")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--spheres", type=int, default=32)
    ap.add_argument("--aa", type=int, default=1)
    ap.add_argument("--no-cull", action="store_true")
    ap.add_argument("--workload", default="c4", choices=["c4", "c2", "c3", "c5"],
                    help="c4 (default, the headline): --spheres random spheres at --size^2; c2: 1 sphere Phong 512^2; "
                         "c3: match_mirror scene (2 spheres + square) 1024^2, A=4; c5: 256 orbit scenes of 64^2, A=4")
    ap.add_argument("--cpu-rows", type=int, default=1024, help="rows of the CPU-baseline band (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--equal-bands", action="store_true",
                    help="N > 1, strong scaling: equal row counts per rank instead of bands of equal estimated cost")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the all-reduce / barrier path even at world size 1 (rehearsal of the N > 1 path on one GPU)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the fixed --size^2 image cut into N row bands (BASELINE configs[3]); "
                         "weak = --size rows PER GPU (the image grows to N*size x size rays)")
    return ap.parse_args()


def relaunch_as_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start N ranks (one process per GPU, RCCL) with
    ``torch.distributed.run`` BEFORE this process touches the GPU, relay their output (rank 0 prints the
    JSON line) and exit with their status.  Never falls back to fewer GPUs."""
    import socket
    import subprocess
    have = torch.cuda.device_count()          # counts devices without initialising the GPU
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible; refusing to run on fewer" % (args.gpus, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    raise SystemExit(subprocess.call(cmd, env=env))


class Job(object):
    """Direct C-ABI driver (no autograd in the timed region)."""

    def __init__(self, dev, size, spheres, aa, row0, rows, cull, workload="c4", full_rows=None):
        from rrt import _native as N
        from rrt import synthetic
        self.N = N
        self.L = N.lib()
        B, specular, transpose, jitter_mode, seed = 1, 1, 1, N.JITTER_CENTER, 0
        cam = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0], dtype=np.float32)
        if workload == "c2":
            w2o, mat, light, kinds = synthetic.one_sphere()
        elif workload == "c3":
            w2o, mat, light, kinds = synthetic.match_mirror(True)
            jitter_mode, seed = N.JITTER_HASH, 1
        elif workload == "c5":
            rs = np.random.RandomState(0)
            B = 256
            _, mat, light, kinds = synthetic.orbit_scene()
            ang = rs.uniform(0, 2 * np.pi, B)
            w2o = np.stack([synthetic.orbit_scene((9 * np.cos(t), 9 * np.sin(t), 32.0))[0] for t in ang])
            cam = np.stack([synthetic.orbit_camera(2.5 if b % 2 == 0 else -2.5) for b in range(B)])
            specular, transpose, jitter_mode, seed = 0, 0, N.JITTER_HASH, 1
        else:
            w2o, mat, light, kinds = synthetic.random_spheres(spheres, seed=0)
        spheres = w2o.shape[-3]
        self.B = B
        self.w2o = torch.tensor(w2o, device=dev)
        self.mat = torch.tensor(mat, device=dev)
        self.light = torch.tensor(light, device=dev)
        self.kinds = torch.tensor(kinds, device=dev)
        self.cam = torch.tensor(cam, device=dev)
        self.S, self.size, self.rows, self.row0 = spheres, size, rows, row0
        sc = N.RrtScene()
        sc.S, sc.B = spheres, B
        sc.w2o, sc.w2o_bstride = self.w2o.data_ptr(), (spheres * 12 if B > 1 else 0)
        sc.material, sc.material_bstride = self.mat.data_ptr(), 0
        sc.light, sc.kind = self.light.data_ptr(), self.kinds.data_ptr()
        self.sc = sc

        def view(flags):
            vw = N.RrtView()
            # root camera (transpose = 1): image axis 0 runs over x_dims, axis 1 over y_dims
            vw.x_dims, vw.y_dims = (full_rows or size), size
            vw.antialias_samples = aa
            vw.row0, vw.rows = row0, rows
            vw.shader_id, vw.specular, vw.transpose = N.SHADER_PHONG, specular, transpose
            vw.jitter_mode, vw.seed = jitter_mode, seed
            vw.flags = flags | (0 if cull else N.FLAG_NO_CULL)
            vw.max_depth = 1.0
            vw.cam, vw.cam_bstride, vw.jitter = self.cam.data_ptr(), (16 if B > 1 else 0), None
            return vw
        self.v_fwd = view(0)
        self.v_fwd_reuse = view(N.FLAG_REUSE_SETUP)
        self.v_bwd = view(N.FLAG_REUSE_SETUP)      # the packed scene of the step's forward is still valid
        self.v_bwd_only = view(N.FLAG_REUSE_SETUP | N.FLAG_SKIP_FINALIZE)   # k_bwd alone (kernel timing)
        nbytes = self.L.rrt_workspace_bytes(ctypes.byref(sc), ctypes.byref(self.v_fwd))
        assert nbytes > 0
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.image = torch.empty((B, rows, size, 3), dtype=torch.float32, device=dev)
        self.grad = torch.empty_like(self.image)
        self.flat = torch.zeros(B * spheres * 12 + spheres * 7 + 6, dtype=torch.float32, device=dev)
        self.stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._p = lambda t: ctypes.c_void_p(t.data_ptr())

    def fwd(self, reuse=False):
        v = self.v_fwd_reuse if reuse else self.v_fwd
        rc = self.L.rrt_render_fwd(ctypes.byref(self.sc), ctypes.byref(v), self._p(self.image), self._p(self.ws),
                                   self.ws.numel(), self.stream)
        self.N.check(rc, "rrt_render_fwd")

    def bwd(self, kernel_only=False):
        S, B = self.S, self.B
        base = self.flat.data_ptr()
        v = self.v_bwd_only if kernel_only else self.v_bwd
        rc = self.L.rrt_render_bwd(ctypes.byref(self.sc), ctypes.byref(v), self._p(self.grad),
                                   ctypes.c_void_p(base), ctypes.c_void_p(base + 4 * B * S * 12),
                                   ctypes.c_void_p(base + 4 * (B * S * 12 + S * 7)), self._p(self.ws), self.ws.numel(),
                                   self.stream)
        self.N.check(rc, "rrt_render_bwd")


def probe_coverage(dev, args, probe=128):
    """Per-row coverage (fraction of pixels that hit a shape) of a probe x probe rendering of the workload's view."""
    pj = Job(dev, probe, args.spheres, 1, 0, probe, not args.no_cull, args.workload)
    pj.fwd()
    torch.cuda.synchronize()
    return (pj.image[0].abs().sum(dim=-1) > 0).float().mean(dim=-1).cpu().numpy()


def time_band(dev, args, full_rows, row0, rows, steps=60):
    """us per step (forward + backward, no collective) of one row band, by the host clock around a synchronised loop"""
    job = Job(dev, args.size, args.spheres, args.aa, row0, rows, not args.no_cull, args.workload, full_rows=full_rows)
    job.fwd()
    job.grad.copy_(2.0 * (job.image - torch.flip(job.image, dims=[2])))
    for _ in range(30):
        job.fwd()
        job.bwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.fwd()
        job.bwd()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def balanced_bands_of(dev, args, full_rows, rank, world, dist, rounds=3, probe=probe_coverage, timer=time_band):
    """Row bands of equal cost for the strong-scaling run, found before anything is timed: a first cut from the coverage
    of a 128-row probe render, then `rounds` measurements -- every rank times its own band, the times are all-gathered,
    rrt.dist.rebalance_bands cuts again -- and the measured partition with the fastest slowest rank is kept.  Every
    decision is taken on all-gathered numbers (rank 0's first cut is broadcast), so all ranks agree by construction
    and run the same number of rounds.  (`probe` / `timer`: the CPU test of this choreography injects synthetic ones.)"""
    from rrt.dist import balanced_bands, rebalance_bands, row_cost_from_coverage
    bands = balanced_bands(row_cost_from_coverage(probe(dev, args), full_rows), world)
    t = torch.tensor(bands, dtype=torch.int64, device=dev)
    dist.broadcast(t, src=0)
    bands = [(int(a), int(b)) for a, b in t.cpu().tolist()]
    best = None
    for _ in range(rounds):
        assert sum(b for _, b in bands) == full_rows and all(bands[r][0] + bands[r][1] == bands[r + 1][0] for r in range(world - 1))
        mine = torch.tensor([timer(dev, args, full_rows, *bands[rank])], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        times = [float(x.item()) for x in every]
        if best is None or max(times) < best[0]:
            best = (max(times), bands, times)
        bands = rebalance_bands(bands, times)
        if bands == best[1]:
            break
    return best[1], best[2]


def cpu_baseline(size, spheres, aa, rows):
    """The CPU oracle (a float32 op-for-op restatement of the reference's Theano graph + autograd,
    array-at-a-time like the Theano graph) on a band of the same workload, timed on this host:
    once on the GPU box's core share (16 threads) and once on ONE thread (Theano's elementwise
    loops are single-threaded without OpenMP, README.md:11) on a band a quarter as high.  The
    Mrays/s of a band is what the 4096-row image would run at (every row costs the same: the
    oracle shades every shape at every pixel); it is an extrapolation, and says so."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import rrt_oracle as O
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.random_spheres(spheres, seed=0)

    def run(rows, threads):
        row0 = max(0, size // 2 - rows // 2)
        rows = min(rows, size - row0)
        G = np.random.RandomState(0).randn(rows, size, 3).astype(np.float32)
        torch.set_num_threads(threads)
        t0 = time.time()
        O.render_and_grads(w2o, mat, light, list(kinds), grad_image=G, x_dims=size, y_dims=size,
                           antialias_samples=aa, row0=row0, rows=rows, dtype=torch.float32)
        dt = time.time() - t0
        return rows * size * aa / dt / 1e6, row0, rows, dt
    # the GPU box gives one GPU a 16-core share; torch's default (all 128 logical CPUs) only adds overhead
    threads = min(16, os.cpu_count() or 1)
    v_all, r0, nr, dt_all = run(rows, threads)
    v_one, r1, n1, dt_one = run(max(16, rows // 4), 1)
    torch.set_num_threads(threads)
    return {"value": v_all, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "rows [%d,%d) of the %dx%d/%d-sphere workload, fwd+bwd, float32 torch-CPU oracle, %.1f s; the figure "
                      "for the whole %d-row image is this band's rate (extrapolation: every row costs the same)"
                      % (r0, r0 + nr, size, size, spheres, dt_all, size),
            "one_thread": {"value": v_one, "unit": "Mrays/s", "cores": 1,
                           "sample": "rows [%d,%d), %.1f s (same extrapolation)" % (r1, r1 + n1, dt_one)}}


def parity_sample(dev, size, spheres, aa, rows=48):
    """max|d grad| of the HIP path vs the float64 oracle (the second half of
    BASELINE.json's metric) on a band of the benchmark workload: the oracle's
    array-at-a-time autograd cannot do the whole 4096^2 image in bench time.
    The oracle is used here as the checker only."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import rrt
    import rrt_oracle as O
    from rrt import synthetic
    w2o, mat, light, kinds = synthetic.random_spheres(spheres, seed=0)
    row0 = max(0, size // 2 - rows // 2)
    rows = min(rows, size - row0)
    G = np.random.RandomState(1).randn(rows, size, 3)
    tw = torch.tensor(w2o, device=dev, requires_grad=True)
    tm = torch.tensor(mat, device=dev, requires_grad=True)
    tl = torch.tensor(light, device=dev, requires_grad=True)
    img = rrt.render(tw, tm, tl, kinds, x_dims=size, y_dims=size, antialias_samples=aa, jitter="center",
                     row0=row0, rows=rows)
    (img * torch.tensor(G, dtype=torch.float32, device=dev)).sum().backward()
    torch.cuda.synchronize()
    oi, ow, om, ol = O.render_and_grads(w2o, mat, light, list(kinds), grad_image=G, x_dims=size, y_dims=size,
                                        antialias_samples=aa, row0=row0, rows=rows, shininess_grad="masked")
    di = (img.detach().cpu().double() - oi).abs().amax(-1)

    def stats(a, b):
        d = (a.detach().cpu().double() - b).abs().max().item()
        return d, d / max(b.abs().max().item(), 1e-30)
    aw, rw = stats(tw.grad, ow)
    am, rm = stats(tm.grad, om)                     # all seven columns, d/dshininess included
    al, rl = stats(tl.grad, ol)
    return {"sample": "rows [%d,%d) of the workload, white-noise upstream gradient, vs the float64 CPU oracle"
                      % (row0, row0 + rows),
            "max_abs_dimage": di.max().item(), "pixels_off_by_more_than_2e-4": int((di > 2e-4).sum().item()),
            "max_abs_dgrad": {"w2o": aw, "material": am, "light": al},
            "max_rel_dgrad": {"w2o": rw, "material": rm, "light": rl}}


def time_kernels(job, n):
    """median duration (s) of k_fwd and of k_bwd over n back-to-back launches of each"""
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
    job.fwd()
    torch.cuda.synchronize()
    for i in range(n):
        ev[i][0].record()
        job.fwd(reuse=True)              # k_fwd alone (packed scene + tile lists reused)
        ev[i][1].record()
        job.bwd(kernel_only=True)        # k_bwd alone (RRT_FLAG_SKIP_FINALIZE)
        ev[i][2].record()
    job.fwd()                            # re-pack: rewinds the accumulators the kernel-only calls left dirty
    torch.cuda.synchronize()
    t_f = float(np.median([e[0].elapsed_time(e[1]) for e in ev])) * 1e-3
    t_b = float(np.median([e[1].elapsed_time(e[2]) for e in ev])) * 1e-3
    return t_f, t_b


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        relaunch_as_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d: they must agree" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path is HIP only (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from rrt.dist import band
    if args.workload == "c2":
        args.size, args.aa = 512, 1
    elif args.workload == "c3":
        args.size, args.aa = 1024, 4
    elif args.workload == "c5":
        args.size, args.aa = 64, 4
    # strong: the size x size image in `world` row bands; weak: `size` rows per rank of a (world*size) x size image
    full_rows = args.size * world if args.scaling == "weak" else args.size
    row0, rows = band(full_rows, rank, world)
    all_bands = None
    if use_dist and args.scaling == "strong" and args.workload == "c4" and not args.equal_bands:
        # Strong scaling: a step is as slow as its slowest rank, and equal row counts differ by up to 20 % in time (N = 4:
        # 33.4 ... 40.1 us).  Bands of equal cost are found here, before anything is timed (balanced_bands_of).
        all_bands, band_times = balanced_bands_of(dev, args, full_rows, rank, world, dist)
        row0, rows = all_bands[rank]
    job = Job(dev, args.size, args.spheres, args.aa, row0, rows, not args.no_cull, args.workload, full_rows=full_rows)
    args.spheres = job.S

    # upstream gradient of the match_mirror loss, computed once: g = 2 (image - fliplr(image0))
    job.fwd()
    job.grad.copy_(2.0 * (job.image - torch.flip(job.image, dims=[2])))
    torch.cuda.synchronize()

    def step():
        job.fwd()
        job.bwd()
        if use_dist:
            dist.all_reduce(job.flat)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed spin-up: a GPU that has been idle runs its first ~30 ms of work 4-5 % slower (clocks), and the timed
    # region may be as short as 2 ms (--steps 20).  Then the W warmup steps, then exactly K timed steps.
    t_spin = time.perf_counter()
    while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
        for _ in range(10):               # no collective in here: the ranks' loop counts differ
            job.fwd()
            job.bwd()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    rays = job.B * full_rows * args.size * args.aa
    value = rays / (elapsed / args.steps) / 1e6

    roofline = None
    kernels = None
    if not args.no_kernel_timing:
        # per-kernel durations: HIP events on the launch stream (torch's current stream), 100 launches of each kernel
        t_f, t_b = time_kernels(job, 100)
        px = job.B * rows * args.size
        kernels = {"k_fwd_us": t_f * 1e6, "k_bwd_us": t_b * 1e6,
                   "k_fwd_GBs": px * BYTES_PER_PIXEL_FWD / t_f / 1e9,
                   "k_bwd_GBs": px * BYTES_PER_PIXEL_BWD / t_b / 1e9,
                   "fwd_plus_bwd_GBs": px * (BYTES_PER_PIXEL_FWD + BYTES_PER_PIXEL_BWD) / (t_f + t_b) / 1e9}
        # counters of the same command (rocprofv3 --pmc, tools/profile_round.sh -> profiles/): HBM bytes and VALU
        # wave-instructions per launch; VALU issue peak from tools/ubench/issue_probe2.hip
        prof = None
        import glob
        cands = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_counters.json")))     # the latest round's
        ppath = cands[-1] if cands else ""
        if ppath and world == 1:
            pj = json.load(open(ppath))
            pw = pj["workload"]
            if (pw["size"], pw["spheres"], pw["aa"]) == (args.size, args.spheres, args.aa):
                prof = pj

        def roof(name, t, alg_bytes):
            """both rooflines of one kernel; `bound` = the wall it is nearer to (by real traffic when known)"""
            hbm_alg = alg_bytes / t / 1e9
            r = {"kernel": name, "us": t * 1e6, "achieved_hbm_fraction": hbm_alg / HBM_PEAK_GBS,
                 "algorithmic_GBs": hbm_alg, "achieved_valu_fraction": None, "traffic": None, "traffic_source": None}
            hbm_real = hbm_alg
            if prof is not None:
                k = prof[name]
                r["traffic"] = k["hbm_bytes"]
                r["traffic_source"] = "from_profile: profiles/%s (%s)" % (os.path.basename(ppath), prof["source"])
                # rocprofv3's own begin-to-end duration of the dispatch; `us` (HIP events) also holds the launch gap
                r["us_kernel_trace_from_profile"] = k["avg_us_kernel_trace"]
                hbm_real = k["hbm_bytes"] / t / 1e9
                peak = prof["valu_peak_inst_per_cycle_per_simd"] * prof["simds"] * prof["clock_ghz"]      # Ginst/s
                r["valu_Ginst_s"] = k["valu_insts"] / t / 1e9
                r["valu_peak_Ginst_s"] = peak
                r["achieved_valu_fraction"] = r["valu_Ginst_s"] / peak
            vf = r["achieved_valu_fraction"] or 0.0
            if vf > hbm_real / HBM_PEAK_GBS:
                r.update({"bound": "valu", "achieved": r["valu_Ginst_s"], "peak": r["valu_peak_Ginst_s"],
                          "unit": "Ginst/s (wave64 VALU instructions)", "frac": vf})
            else:
                r.update({"bound": "hbm", "achieved": hbm_alg, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": hbm_alg / HBM_PEAK_GBS})
            return r
        r_f = roof("k_fwd", t_f, px * BYTES_PER_PIXEL_FWD)
        r_b = roof("k_bwd", t_b, px * BYTES_PER_PIXEL_BWD)
        roofline, roofline_other = (r_f, r_b) if t_f >= t_b else (r_b, r_f)
        kernels["step_algorithmic_GBs"] = px * (BYTES_PER_PIXEL_FWD + BYTES_PER_PIXEL_BWD) / (elapsed / args.steps) / 1e9
        kernels["step_hbm_fraction"] = kernels["step_algorithmic_GBs"] / HBM_PEAK_GBS
    else:
        roofline_other = None

    if rank == 0:
        cpu = None
        parity = None
        if world == 1 and args.cpu_rows > 0 and args.workload == "c4":
            cpu = cpu_baseline(args.size, args.spheres, args.aa, args.cpu_rows)
            parity = parity_sample(dev, args.size, args.spheres, args.aa)
        out = {
            "metric": "Mrays/s fwd+bwd at %dx%d/%d spheres" % (full_rows, args.size, args.spheres),
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": {"c4": "%d random spheres (seed 0), %dx%d, antialias_samples=%d, Phong+specular, "
                                          "root camera, dense upstream grad; %d row band(s), one per GPU: %s"
                                          % (args.spheres, full_rows, args.size, args.aa, world,
                                             ("of equal measured cost (probe render + 3 timed rounds before the run), rows " + "/".join(str(b) for _, b in all_bands))
                                             if all_bands else "of %d rows" % rows),
                                    "c2": "config 2: 1 sphere Phong, 512x512, antialias_samples=1",
                                    "c3": "config 3: match_mirror.py scene (2 spheres + square), 1024x1024, "
                                          "antialias_samples=4, hash jitter",
                                    "c5": "config 5: 256 orbit scenes (2 spheres, stereo cameras, no specular), 64x64, "
                                          "antialias_samples=4, one batched launch"}[args.workload],
                       "cull": not args.no_cull, "parallelism": "rowband%d" % world, "spinup_ms": args.spinup_ms},
            "roofline": roofline, "roofline_other_kernel": roofline_other, "kernels": kernels, "cpu_baseline": cpu,
            "parity": parity,
        }
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
