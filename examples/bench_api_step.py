#!/usr/bin/env python3
"""Time of ONE gradient-descent step through the reference-style class API
(Scene/Sphere/translate/... rebuilt every step, autograd, in-place update),
eager versus captured in a HIP graph (GDOptimizer.optimize(graph=True)).
Needs an MI355X.   python examples/bench_api_step.py [--size 1024] [--spheres 8]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "reversible-raytracer_amd"))
import rrt                                                     # noqa: E402
from rrt import (Camera, GDOptimizer, Light, Material, PhongShader, Scene, Sphere, scale, shared,   # noqa: E402
                 translate)

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--spheres", type=int, default=8)
ap.add_argument("--steps", type=int, default=50)
args = ap.parse_args()
rs = np.random.RandomState(0)
centres, mats, radii = [], [], []
for _ in range(args.spheres):
    z = rs.uniform(4, 12)
    centres.append(shared(np.array([rs.uniform(-.4, .4) * z, rs.uniform(-.4, .4) * z, z], dtype="float32")))
    radii.append(float(rs.uniform(.15, .45)))
    mats.append(Material(tuple(rs.uniform(.1, .9, 3)), 0.3, float(rs.uniform(.6, .9)), float(rs.uniform(.3, .5)), 50.))
light = Light((-1., -1., 2.), (0.961, 1., 0.87))
cam = Camera(args.size, args.size)


def build():
    shapes = [Sphere(translate(c) * scale((r, r, r)), m) for c, r, m in zip(centres, radii, mats)]
    return Scene(shapes, [light], cam, PhongShader()).build(jitter="center")


target = torch.flip(build().detach(), dims=[1])
cost = lambda: ((build() - target) ** 2).sum()

# the same scene through the vectorised front end (rrt.batched): O(1) tensor ops per step
from rrt import batched                                        # noqa: E402
dev = centres[0].device
C = torch.stack([c.detach() for c in centres]).requires_grad_(True)
R = torch.tensor(radii, device=dev)
M = torch.stack([m.as_row() for m in mats]).detach()
Lrow = light.as_row().detach()


def build_v():
    return rrt.render(batched.w2o_translate_scale(C, R), M, Lrow, x_dims=args.size, y_dims=args.size,
                      antialias_samples=4, jitter="center")


cost_v = lambda: ((build_v() - target) ** 2).sum()
assert torch.allclose(build_v(), build(), atol=1e-6)

for graph in (False, True):
    train = GDOptimizer().optimize(centres, cost, 1e-9, 0.1, graph=graph)
    for _ in range(5):
        train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        train()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print("%-6s  %.3f ms/step  (%d spheres, %dx%d)" % ("graph" if graph else "eager", dt * 1e3, args.spheres,
                                                        args.size, args.size))
for graph in (False, True):
    train = GDOptimizer().optimize([C], cost_v, 1e-9, 0.1, graph=graph)
    for _ in range(5):
        train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        train()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print("%-6s  %.3f ms/step  vectorised front end (rrt.batched)" % ("graph" if graph else "eager", dt * 1e3))
