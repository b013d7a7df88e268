#!/usr/bin/env python3
"""How much host time does one distributed step cost?  On ONE GPU with a world-size-1 RCCL group: a tiny image
(GPU time small) so that the loop is host-bound, with and without the all_reduce of the flat gradient buffer,
eager and captured in a HIP graph.  Tells whether `bench.py --gpus 8` (512-row bands, ~35 us of GPU work per
rank and step) would be bound by the Python loop.
    python tools/host_overhead_probe.py"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "reversible-raytracer_amd"))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for size, rows in ((256, 256), (4096, 512)):
    job = bench.Job(dev, size, 32, 1, 0, rows, True)
    job.fwd()
    job.grad.copy_(2.0 * (job.image - torch.flip(job.image, dims=[2])))

    def step(ar):
        job.fwd()
        job.bwd()
        if ar:
            dist.all_reduce(job.flat)

    def timeit(fn, n=300):
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    res = {"size": size, "rows": rows, "eager_us": timeit(lambda: step(False)), "eager_allreduce_us": timeit(lambda: step(True))}
    # host-only cost of issuing a step (no synchronisation inside the loop is what the numbers above already are);
    # the same step captured in a HIP graph
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step(True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        job.stream = __import__("ctypes").c_void_p(side.cuda_stream)
        with torch.cuda.graph(g, stream=side):
            job.stream = __import__("ctypes").c_void_p(torch.cuda.current_stream().cuda_stream)
            step(True)
        res["graph_allreduce_us"] = timeit(g.replay)
    except Exception as e:   # noqa: BLE001
        res["graph_error"] = repr(e)[:200]
    print(res)
dist.destroy_process_group()
