#!/usr/bin/env python3
"""One seed of tools/soak_random_scenes.py (GPU box): python tools/soak_one.py SEED CENTRE(0|1)   [RRT_TUNE_* in the environment]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "reversible-raytracer_amd"), os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch
import test_gpu_parity as T
try:
    T._random_scene_case(torch.device("cuda:0"), int(sys.argv[1]), bool(int(sys.argv[2])))
    print("seed", sys.argv[1], "ok")
except AssertionError as e:
    print("seed", sys.argv[1], "FAILED", str(e)[:300])
