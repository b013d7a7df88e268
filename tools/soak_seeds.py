#!/usr/bin/env python3
"""Run given (seed, centre) cases of the random-scene parity check and print pass / fail (RRT_LIB selects the library)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "reversible-raytracer_amd"), os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch
import test_gpu_parity as T
dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    seed, centre = spec.split(":")
    try:
        T._random_scene_case(dev, int(seed), centre == "1")
        print(spec, "ok")
    except AssertionError as e:
        print(spec, "FAIL", str(e)[:200])
