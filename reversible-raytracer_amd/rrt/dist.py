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


# Cost of a row band on one MI355X, fitted to the bands of the 4096^2 / 32-sphere workload (tools/band_cost_model.py):
#   step(us) = 24.3 + 25.6 * rows / 4096 + 91.4 * covered_rows / 4096,   covered_rows = sum over the band's rows of the
# fraction of the row's pixels that hit a shape.  The fixed part is every rank's; the rest is what a partition can even out.
ROW_COST_COVERED = 91.4 / 25.6


def row_cost_from_coverage(coverage, n_rows):
    """Relative cost of each of `n_rows` image rows from the per-row coverage (fraction of pixels that hit a shape, in
    [0, 1]) of ANY rendering of the same view -- typically a cheap low-resolution probe, stretched to n_rows."""
    import numpy as np
    cov = np.clip(np.asarray(coverage, dtype=np.float64).reshape(-1), 0.0, 1.0)
    if cov.size == 0:
        return np.ones(n_rows)
    idx = np.minimum((np.arange(n_rows) * cov.size) // max(n_rows, 1), cov.size - 1)
    return 1.0 + ROW_COST_COVERED * cov[idx]


def balanced_bands(row_cost, world_size, align=4):
    """Contiguous row bands [(row0, rows)] * world_size of about equal summed cost (strong scaling: a step is as slow as
    its slowest rank, and an image's rows differ in what they contain).  Boundaries fall on multiples of `align` rows
    (the kernels' tile height) where the image is large enough; every rank gets at least one row while rows last.
    Deterministic: every rank computes the same partition from the same costs."""
    import numpy as np
    cost = np.maximum(np.asarray(row_cost, dtype=np.float64).reshape(-1), 0.0)
    n = cost.size
    if world_size <= 1 or n == 0:
        return [(0, n)] + [(n, 0)] * (world_size - 1)
    if n < world_size * max(align, 1) or not np.isfinite(cost).all() or cost.sum() <= 0.0:
        return [band(n, r, world_size) for r in range(world_size)]
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = [0]
    for r in range(1, world_size):
        c = int(np.searchsorted(cum, cum[-1] * r / world_size))
        c = int(round(c / align)) * align
        lo, hi = cuts[-1] + align, n - (world_size - r) * align     # room for every rank on either side
        cuts.append(min(max(c, lo), hi))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world_size)]


def rebalance_bands(bands, step_us, fixed_us=24.0, align=4):
    """One round of measured-cost balancing: `bands` = the current [(row0, rows)] per rank, `step_us` = every rank's
    measured time per step on its band (all-gathered, so every rank passes the same numbers and gets the same answer).
    A rank's time above the fixed per-launch cost is spread evenly over its rows -- rows with several overlapping shapes
    cost a multiple of what their coverage says, which only a measurement sees -- and the image is cut again into bands
    of equal estimated time."""
    import numpy as np
    n = sum(b for _, b in bands)
    cost = np.ones(n)
    for (a, b), t in zip(bands, step_us):
        if b > 0:
            cost[a:a + b] = max(float(t) - fixed_us, 0.05 * float(t), 1e-3) / b
    return balanced_bands(cost, len(bands), align)


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
