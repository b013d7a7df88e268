#!/usr/bin/env python3
"""The reference's match_mirror.py driver on the HIP path: two spheres + a square,
loss = sum((image - fliplr(initial render))^2), 90 gradient-descent steps on the
two sphere centres (match_mirror.py:16-53).  Needs an MI355X.

    python examples/match_mirror.py [--steps 90] [--size 128] [--out output]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "reversible-raytracer_amd"))
from rrt.optimize import GDOptimizer          # noqa: E402
from rrt.scene import *                       # noqa: E402,F401,F403
from rrt.shader import *                      # noqa: E402,F401,F403
from rrt.transform import *                   # noqa: E402,F401,F403
from rrt.util import draw                     # noqa: E402
from rrt import shared                        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=90)
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--out", default="output")
args = ap.parse_args()
os.makedirs(args.out, exist_ok=True)

center1 = shared(np.asarray([-.5, -.5, 4], dtype="float32"))
center2 = shared(np.asarray([.5, .5, 4], dtype="float32"))
material1 = Material((0.2, 0.9, 0.4), 0.3, 0.7, 0.5, 50.)
material2 = Material((0.87, 0.1, 0.507), 0.3, 0.9, 0.4, 50.)
light = Light((-1., -1., 2.), (1., 0.87, 0.961))
camera = Camera(args.size, args.size)
shader = PhongShader()
seed = 1234                                   # the reference bakes ONE unseeded jitter draw into its graph


def build():
    objs = [Sphere(translate(center1), material1), Sphere(translate(center2), material2),
            Square(translate((0, 0, 3)) * rotate(50, [0., 1., 0.]), material2)]
    return Scene(objs, [light], camera, shader).build(jitter=seed)


print("Rendering initial scene")
render = build().detach()
flipped = torch.flip(render, dims=[1])        # np.fliplr
draw(os.path.join(args.out, "0.png"), render)
draw(os.path.join(args.out, "0lr.png"), flipped)

cost = lambda: ((build() - flipped) ** 2).sum()
print("Building gradient functions")
train = GDOptimizer().optimize([center1, center2], cost, 0.000008, 0.1)
for i in range(args.steps):
    print("Step", i + 1)
    print(train())
    draw(os.path.join(args.out, "%d.png" % (i + 1)), build())
print("centres:", center1.detach().cpu().numpy(), center2.detach().cpu().numpy())
