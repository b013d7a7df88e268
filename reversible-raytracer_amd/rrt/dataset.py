"""Dataset writers / readers in the reference's on-disk formats (the data
format either side of the render path).

* ``generate_data.py:44-52``      ``np.savez('dataset', dataset)`` with
  ``dataset[i] = (render[:, :, 0] * 255).astype(uint8)`` -> ``arr_0`` uint8 [n, x, y]
* ``orbit_experiments/planet_orbit.py:55-67``  ``arr_0`` uint8 [n, 2, x, y, 3]
  (two cameras) and the target centres; loaders divide by 255.0
  (``orbit_experiments/test_optimization.py:56,109``).

The renders come from ``Scene.build()`` (the HIP path); this module only does
the uint8 quantisation and the ``.npz`` layout.
"""
import numpy as np
import torch


def quantize(render):
    """(render * 255).astype(np.uint8) exactly as the reference (truncation, no
    clipping: callers render in [0,1])."""
    r = render.detach().cpu().numpy() if torch.is_tensor(render) else np.asarray(render)
    return (r * 255).astype(np.uint8)


def save_depth_dataset(path, renders):
    """generate_data.py: n single-channel renders -> arr_0 uint8 [n, x, y]."""
    data = np.stack([quantize(r)[:, :, 0] for r in renders])
    np.savez(path, data)
    return data


def save_orbit_dataset(path, stereo_renders, targets=None, target_path=None):
    """planet_orbit.py: n (left, right) render pairs -> arr_0 uint8 [n, 2, x, y, 3].
    ``targets`` [n, 3]: saved whole to ``target_path`` (the reference saves only
    the last one, planet_orbit.py:67 -- a bug we do not reproduce)."""
    data = np.stack([np.stack([quantize(l), quantize(r)]) for l, r in stereo_renders])
    np.savez(path, data)
    if targets is not None and target_path is not None:
        np.savez(target_path, np.asarray(targets, dtype=np.float64))
    return data


def load_dataset(path):
    """``np.load(path)['arr_0'] / 255.0`` as float32 (test_optimization.py:56-57)."""
    return (np.load(path)["arr_0"] / 255.0).astype("float32")
