#!/usr/bin/env python3
"""orbit_experiments/planet_orbit.py on the HIP path, batched: n random orbit
positions x 2 cameras rendered by ONE launch (batch axis of rrt.render), written
in the reference's .npz layout (arr_0 uint8 [n,2,64,64,3]).  Needs an MI355X.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "reversible-raytracer_amd"))
import rrt                                    # noqa: E402
from rrt import dataset, synthetic            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=100)
ap.add_argument("--out", default="orbit_dataset")
args = ap.parse_args()
rs = np.random.RandomState(0)
dev = torch.device("cuda:0")

targets = []
for _ in range(args.n):                       # planet_orbit.py:44-53
    x = (1 if rs.rand() > 0.5 else -1) * rs.rand()
    y = (1 if rs.rand() > 0.5 else -1) * np.sqrt(1.0 - x ** 2)
    targets.append((x * 9, y * 9, 32.0))
w = np.stack([synthetic.orbit_scene(t)[0] for t in targets for _ in range(2)])          # [2n, 2, 3, 4]
_, mat, light, kinds = synthetic.orbit_scene()
cams = np.stack([synthetic.orbit_camera(y) for _ in targets for y in (2.5, -2.5)])      # [2n, 16]
img = rrt.render(torch.tensor(w, device=dev), torch.tensor(mat, device=dev), torch.tensor(light, device=dev),
                 kinds, torch.tensor(cams, device=dev), x_dims=64, y_dims=64, antialias_samples=4,
                 specular=False, jitter=7)
img = img.view(args.n, 2, 64, 64, 3).cpu()
data = dataset.save_orbit_dataset(args.out, [(img[i, 0], img[i, 1]) for i in range(args.n)], targets,
                                  args.out + "_target")
print("wrote", args.out + ".npz", data.shape, data.dtype)
