"""Row-band sharding of one render over the GPUs of a node.

Every pixel is independent in the forward pass; the backward pass needs one
tiny sum.  Rank r renders image rows [r*H/G, (r+1)*H/G) (``row0/rows`` of
``rrt_view``) with the full (replicated) scene parameters; the backward
kernels leave [g_w2o | g_material | g_light] in ONE flat float32 buffer
(<= 2.5 KB at 32 shapes) which is summed across ranks with a single
``all_reduce`` -- RCCL over xGMI when the process group's backend is "nccl",
latency-bound at this size, so one fused buffer and one collective.  The loss
must be pixel-separable (true for every loss in the reference:
optimize_brightness.py:51, match_mirror.py:45, autoencoder_2ly.py:91).

The reference has no counterpart: it is single-process.
"""
import contextlib

import torch
import torch.distributed as dist

from .render import _grad_hook   # the list RenderFn.backward consults (rrt.render the module is shadowed by rrt.render the function)


def band(n_rows, rank, world_size):
    """Contiguous row band of rank `rank`: (row0, rows); the first
    n_rows % world_size ranks get one extra row."""
    base, extra = divmod(n_rows, world_size)
    row0 = rank * base + min(rank, extra)
    rows = base + (1 if rank < extra else 0)
    return row0, rows


def all_reduce_flat(flat, group=None):
    """Sum the flat parameter-gradient buffer over the ranks, in place (a no-op
    without a process group; with one, the collective runs even at world size 1
    so that a single-GPU rehearsal exercises the same RCCL path)."""
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


@contextlib.contextmanager
def sharded(group=None):
    """Inside this context every RenderFn.backward all-reduces its flat
    gradient buffer over `group` before handing the pieces to autograd, so all
    ranks see the gradient of the WHOLE image and take identical steps."""
    prev = _grad_hook[0]
    _grad_hook[0] = lambda flat: all_reduce_flat(flat, group)
    try:
        yield
    finally:
        _grad_hook[0] = prev


def render_band(scene, antialias_samples=4, jitter=None, group=None):
    """This rank's band of ``scene.build()``: (image_band, row0, rows)."""
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    rk = dist.get_rank(group) if dist.is_initialized() else 0
    full = scene.camera.x_dims if scene.camera.transpose else scene.camera.y_dims
    row0, rows = band(full, rk, ws)
    return scene.build(antialias_samples, jitter=jitter, row0=row0, rows=rows), row0, rows


def gather_image(image_band, n_rows, group=None):
    """Optional: assemble the full image on every rank (only needed to save
    frames; 201 MB at 4096^2).  Bands may differ by one row, so pad + trim."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return image_band
    ws = dist.get_world_size(group)
    max_rows = -(-n_rows // ws)
    pad = torch.zeros((max_rows,) + tuple(image_band.shape[1:]), dtype=image_band.dtype,
                      device=image_band.device)
    pad[:image_band.shape[0]] = image_band
    parts = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(parts, pad, group=group)
    out = [parts[r][:band(n_rows, r, ws)[1]] for r in range(ws)]
    return torch.cat(out, dim=0)
