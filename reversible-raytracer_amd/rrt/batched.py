"""Vectorised scene parameters (an extension: the reference builds one
``Transform`` object per shape, which in eager PyTorch costs ~25 tiny kernels
per shape and step; with hundreds of shapes that plumbing dwarfs the render
kernels even inside a HIP graph).

These helpers produce the ``w2o[S,3,4]`` / ``material[S,7]`` tensors that
``rrt.render`` consumes with a handful of tensor ops, differentiable w.r.t.
their inputs.  They restate the same algebra as ``transform.py``:

    o2w = translate(c) * scale(s)          (transform.py:60-93)
    w2o = o2w.inverse() = scale(1/s) * translate(-c)      (transform.py:31-38)
"""
import torch

from .config import floatX


def w2o_translate_scale(centers, scales):
    """centers [S,3], scales [S] (uniform) or [S,3] -> w2o [S,3,4] =
    rows 0..2 of ``(translate(c) * scale(s)).inverse().m``."""
    centers = torch.as_tensor(centers, dtype=floatX)
    scales = torch.as_tensor(scales, dtype=floatX, device=centers.device)
    if scales.dim() == 1:
        scales = scales.unsqueeze(1).expand(-1, 3)
    inv = 1.0 / scales
    return torch.cat([torch.diag_embed(inv), (-centers * inv).unsqueeze(2)], dim=2)


def materials(colors, ks, kd, ka, shininess):
    """colors [S,3] and per-shape (or scalar) ks, kd, ka, shininess -> material [S,7]
    (the field order of ``Material.as_row``: r,g,b,ks,kd,ka,shininess)."""
    colors = torch.as_tensor(colors, dtype=floatX)
    S = colors.shape[0]
    cols = [torch.as_tensor(v, dtype=floatX, device=colors.device).expand(S).unsqueeze(1) for v in (ks, kd, ka, shininess)]
    return torch.cat([colors] + cols, dim=1)
