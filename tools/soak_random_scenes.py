#!/usr/bin/env python3
"""One-off soak on the GPU box: the random-scene parity case of tests/test_gpu_parity.py for many more seeds than the
suite carries (general affine Spheres and Squares, ragged sizes, 1-3 samples, all shaders, both cameras, hash and
centre jitter), each against the float64 oracle with the suite's own tolerances.
    python tools/soak_random_scenes.py [first_seed] [count]"""
import os, sys, time, traceback
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "reversible-raytracer_amd"), os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch
import test_gpu_parity as T
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
dev = torch.device("cuda:0")
bad = []
t0 = time.time()
for seed in range(first, first + count):
    for centre in (False, True):
        try:
            T._random_scene_case(dev, seed, centre)
        except Exception as e:       # noqa: BLE001
            bad.append((seed, centre, repr(e)[:300]))
            traceback.print_exc()
    if (seed - first) % 10 == 9:
        print("seeds %d..%d done, %d failures, %.0f s" % (first, seed, len(bad), time.time() - t0), flush=True)
print("soak: %d scenes x 2 jitter modes, %d failures" % (count, len(bad)))
for b in bad:
    print("FAILED", b)
sys.exit(1 if bad else 0)
