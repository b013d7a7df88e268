"""Utilities -- reference ``util.py`` (broadcasted_switch :27-28, get_epsilon
:23-24, initialize_weight :10-19, draw :54-55, drawWithMarkers :44-52)."""
import numpy as np
import torch

from .config import device, floatX


def initialize_weight(n_vis, n_hid, W_name, numpy_rng, rng_dist):
    """util.py:10-19 -> a leaf float32 tensor (the reference returns theano.shared)."""
    if "uniform" in rng_dist:
        W = numpy_rng.uniform(low=-np.sqrt(6.0 / (n_vis + n_hid)), high=np.sqrt(6.0 / (n_vis + n_hid)),
                              size=(n_vis, n_hid)).astype(np.float32)
    elif rng_dist == "normal":
        W = 0.01 * numpy_rng.normal(size=(n_vis, n_hid)).astype(np.float32)
    else:
        raise ValueError(rng_dist)
    t = torch.as_tensor(W, dtype=floatX).to(device())
    t.requires_grad_(True)
    return t


def get_epsilon(epsilon, n, i):
    """Decaying learning rate (util.py:23-24)."""
    return float(epsilon / (1 + i / float(n)))


def broadcasted_switch(a, b, c):
    """util.py:27-28: switch on a [H,W] mask broadcast over the channel axis."""
    return torch.where(a.unsqueeze(-1), b, c)


def _to_uint8_stretched(im):
    """scipy.misc.imsave semantics (util.py:7,55): contrast-stretch min..max to
    0..255 -- saved PNGs are NOT absolute intensities."""
    im = np.asarray(im.detach().cpu() if torch.is_tensor(im) else im, dtype=np.float64)
    lo, hi = im.min(), im.max()
    if hi <= lo:
        return np.zeros(im.shape, dtype=np.uint8)
    return ((im - lo) * (255.0 / (hi - lo)) + 0.5).astype(np.uint8)


def draw(fname, im, absolute=False):
    """util.py:54-55.  ``absolute=True`` writes clip(im,0,1)*255 instead of the
    reference's min..max stretch."""
    from PIL import Image
    if absolute:
        a = np.asarray(im.detach().cpu() if torch.is_tensor(im) else im, dtype=np.float64)
        arr = (np.clip(a, 0, 1) * 255).astype(np.uint8)
    else:
        arr = _to_uint8_stretched(im)
    Image.fromarray(arr).save(fname)


def drawWithMarkers(fname, im, markers=((85, 90), (90, 50))):
    """util.py:44-52: the image with 6x6 red boxes around the two probe pixels
    of optimize_brightness.py:51 (boxes are drawn at (x=85,y=90) and (x=90,y=50))."""
    arr = _to_uint8_stretched(im).copy()
    if arr.ndim == 2:
        arr = np.stack([arr] * 3, axis=-1)
    h, w = arr.shape[:2]
    for (x, y) in markers:
        x0, x1, y0, y1 = max(x - 3, 0), min(x + 3, w - 1), max(y - 3, 0), min(y + 3, h - 1)
        arr[y0:y1 + 1, [x0, x1]] = (255, 0, 0)
        arr[[y0, y1], x0:x1 + 1] = (255, 0, 0)
    from PIL import Image
    Image.fromarray(arr).save(fname)
