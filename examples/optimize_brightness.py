#!/usr/bin/env python3
"""The reference's optimize_brightness.py driver on the HIP path: a sphere and a
rotated, scaled ellipsoid; gradient ascent on the brightness of two probe pixels
w.r.t. the two centres (optimize_brightness.py:19-58; the live pdb.set_trace()
of the reference is not reproduced).  Needs an MI355X.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "reversible-raytracer_amd"))
from rrt.optimize import GDOptimizer          # noqa: E402
from rrt.scene import *                       # noqa: E402,F401,F403
from rrt.shader import *                      # noqa: E402,F401,F403
from rrt.shape import *                       # noqa: E402,F401,F403
from rrt.transform import *                   # noqa: E402,F401,F403
from rrt.util import drawWithMarkers          # noqa: E402
from rrt import shared                        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=90)
ap.add_argument("--out", default="output")
args = ap.parse_args()
os.makedirs(args.out, exist_ok=True)

center1 = shared(np.asarray([-.5, -.5, 4], dtype="float32"))
center2 = shared(np.asarray([.5, .5, 4], dtype="float32"))
material1 = Material((0.2, 0.9, 0.4), 0.3, 0.7, 0.5, 50.)
material2 = Material((0.87, 0.1, 0.507), 0.3, 0.9, 0.4, 50.)
light = Light((-1., -1., 2.), (0.961, 1., 0.87))
camera = Camera(128, 128)
shader = PhongShader()


def build():
    t1 = translate(center1)
    t2 = translate(center2) * rotate(90, (0, 0, 1)) * scale((1, 2, 1.5))
    return Scene([Sphere(t1, material1), Sphere(t2, material2)], [light], camera, shader).build(jitter=99)


print("Rendering initial scene")
drawWithMarkers(os.path.join(args.out, "0.png"), build())
print("Building gradient functions")
loss = lambda: -build()[90, 85].sum() - build()[50, 90].sum()
train = GDOptimizer().optimize([center1, center2], loss, 0.0008, 0.1)
for i in range(args.steps):
    print("Step", i + 1)
    print(train())
    drawWithMarkers(os.path.join(args.out, "%d.png" % (i + 1)), build())
