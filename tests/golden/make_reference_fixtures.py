#!/usr/bin/env python3
"""Copies the DATA artifacts the reference's own Theano path produced into
tests/golden/ (run once, in the build container where /root/reference exists).

Nothing here is reference source code: these are output files (uint8 renders,
a float64[3] target, a JPEG) plus the known-answer numbers of
``test/test_transform.py`` which are re-typed as data in
``tests/golden/transform_kat.json``.

  orbit_sample99.npz   arr_0[99] of orbit_experiments/orbit_dataset.npz
                       (2 cameras x 64x64x3 uint8) + orbit_target.npz arr_0
                       (float64[3]); the only dataset sample whose scene
                       parameters are known (planet_orbit.py:66-67 saves the
                       LAST target only)
  match_mirror_0.npy   output/0.jpg decoded with PIL to uint8[128,128,3]
                       (initial render of match_mirror.py:16-39)
"""
import json
import os

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ds = np.load(os.path.join(REF, "orbit_experiments", "orbit_dataset.npz"))["arr_0"]
    tgt = np.load(os.path.join(REF, "orbit_experiments", "orbit_target.npz"))["arr_0"]
    np.savez_compressed(os.path.join(HERE, "orbit_sample99.npz"),
                        images=ds[99], target=tgt)
    im = np.asarray(Image.open(os.path.join(REF, "output", "0.jpg")).convert("RGB"))
    np.save(os.path.join(HERE, "match_mirror_0.npy"), im)
    kat = {
        "source": "test/test_transform.py:9-41",
        "rotate_20_z": [[0.93969262, -0.34202015, 0.0, 0.0],
                        [0.34202015, 0.93969262, 0.0, 0.0],
                        [0.0, 0.0, 1.0, 0.0],
                        [0.0, 0.0, 0.0, 1.0]],
        "translate_456_rotate_20_z": [[0.93969262, -0.34202015, 0.0, 4.0],
                                      [0.34202015, 0.93969262, 0.0, 5.0],
                                      [0.0, 0.0, 1.0, 6.0],
                                      [0.0, 0.0, 0.0, 1.0]],
        "apply": {"transform": "translate((4,5,6))*rotate(90,(0,0,1))",
                  "origin_in": [1, 0, 0], "ray_in": [0, 1, 0], "grid": [10, 10],
                  "origin_out": [4, 6, 6], "ray_out": [-1, 0, 0]},
    }
    with open(os.path.join(HERE, "transform_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
