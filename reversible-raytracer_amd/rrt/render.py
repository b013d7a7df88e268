"""The render operator: ``Scene.build()`` and its ``T.grad`` as one
``torch.autograd.Function`` over the HIP kernels (through the C ABI of
``include/rrt.h``; PyTorch only supplies device memory and the stream).

Replaces, for the hot path, the Theano graph the reference builds in
``scene.py:18-52`` (forward) and differentiates in ``optimize.py:25``
(backward).  Tensors must live on a ROCm device; there is no CPU path.
"""
import ctypes

import numpy as np
import torch

from . import _native as N

_IDENT_CAM = (1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 1.0, 0)


class RenderOptions(object):
    """Everything about a render that is not a differentiable tensor."""

    __slots__ = ("x_dims", "y_dims", "antialias_samples", "shader", "specular", "max_depth",
                 "transpose", "jitter_mode", "seed", "row0", "rows", "cull", "shadows")

    def __init__(self, x_dims, y_dims, antialias_samples=4, shader="phong", specular=True,
                 max_depth=1.0, transpose=True, jitter_mode=N.JITTER_CENTER, seed=0,
                 row0=0, rows=None, cull=True, shadows=False):
        self.x_dims = int(x_dims)
        self.y_dims = int(y_dims)
        self.antialias_samples = int(antialias_samples)
        self.shader = shader
        self.specular = bool(specular)
        self.max_depth = float(max_depth)
        self.transpose = bool(transpose)
        self.jitter_mode = int(jitter_mode)
        self.seed = int(seed) & 0xFFFFFFFF
        self.row0 = int(row0)
        full = self.x_dims if self.transpose else self.y_dims
        self.rows = full - self.row0 if rows is None else int(rows)
        self.cull = bool(cull)
        self.shadows = bool(shadows)     # extension (RRT_FLAG_SHADOWS): not something the reference renders

    @property
    def ncols(self):
        return self.y_dims if self.transpose else self.x_dims

    @property
    def rows_full(self):
        return self.x_dims if self.transpose else self.y_dims


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


def _jitter_allows_cull(jitter):
    """The tile cull bounds a tile's rays assuming sampleDist in [0, 1] (what the reference's
    (U + sample) / A always is; include/rrt.h).  An explicit jitter tensor outside that range
    would silently drop hit shapes from tiles, so it is range-checked here (one tiny reduction
    + sync; explicit jitter is the parity-run mode) and culling is switched off for it.  Inside a
    HIP-graph capture the check cannot run: cull off."""
    if torch.cuda.is_current_stream_capturing():
        return False
    lo, hi = torch.aminmax(jitter)
    return bool(lo >= 0.0) and bool(hi <= 1.0)


def _structs(w2o, material, light, cam, kinds, jitter, opts, flags, cull=None):
    B = max(w2o.shape[0] if w2o.dim() == 4 else 1,
            material.shape[0] if material.dim() == 3 else 1,
            cam.shape[0] if cam.dim() == 2 else 1)
    S = w2o.shape[-3]
    sc = N.RrtScene()
    sc.S, sc.B = S, B
    sc.w2o = _ptr(w2o).value
    sc.w2o_bstride = S * 12 if (w2o.dim() == 4 and w2o.shape[0] > 1) else 0
    sc.material = _ptr(material).value
    sc.material_bstride = S * 7 if (material.dim() == 3 and material.shape[0] > 1) else 0
    sc.light = _ptr(light).value
    sc.kind = _ptr(kinds).value if kinds is not None else None
    vw = N.RrtView()
    vw.x_dims, vw.y_dims, vw.antialias_samples = opts.x_dims, opts.y_dims, opts.antialias_samples
    vw.row0, vw.rows = opts.row0, opts.rows
    vw.shader_id = N.SHADER_PHONG if opts.shader == "phong" else N.SHADER_DEPTH
    vw.specular = int(opts.specular)
    vw.transpose = int(opts.transpose)
    vw.jitter_mode = opts.jitter_mode
    vw.seed = opts.seed
    vw.flags = flags | (0 if (opts.cull if cull is None else cull) else N.FLAG_NO_CULL) | \
        (N.FLAG_SHADOWS if opts.shadows else 0)
    vw.max_depth = opts.max_depth
    vw.cam = _ptr(cam).value
    vw.cam_bstride = 16 if (cam.dim() == 2 and cam.shape[0] > 1) else 0
    vw.jitter = _ptr(jitter).value if jitter is not None else None
    return sc, vw, S, B


def _check_inputs(w2o, material, light, cam, kinds, jitter, opts):
    dev = w2o.device
    if dev.type != "cuda":
        raise RuntimeError("rrt.render needs tensors on a ROCm GPU (got %s); there is no CPU render path" % dev)
    for name, t in (("material", material), ("light", light), ("cam", cam)):
        if t.device != dev:
            raise RuntimeError("%s is on %s, w2o on %s" % (name, t.device, dev))
    if w2o.shape[-2:] != (3, 4) or w2o.dim() not in (3, 4):
        raise ValueError("w2o must be [S,3,4] or [B,S,3,4], got %s" % (tuple(w2o.shape),))
    S = w2o.shape[-3]
    if material.shape[-2:] != (S, 7) or material.dim() not in (2, 3):
        raise ValueError("material must be [S,7] or [B,S,7] with S=%d, got %s" % (S, tuple(material.shape)))
    if tuple(light.shape) != (6,):
        raise ValueError("light must be [6]")
    if cam.shape[-1] != 16 or cam.dim() not in (1, 2):
        raise ValueError("cam must be [16] or [B,16]")
    bs = set()
    if w2o.dim() == 4:
        bs.add(w2o.shape[0])
    if material.dim() == 3:
        bs.add(material.shape[0])
    if cam.dim() == 2:
        bs.add(cam.shape[0])
    bs.discard(1)
    if len(bs) > 1:
        raise ValueError("inconsistent batch sizes %s" % sorted(bs))
    if kinds is not None and (kinds.dtype != torch.uint8 or tuple(kinds.shape) != (S,)):
        raise ValueError("kinds must be uint8 [S]")
    if opts.jitter_mode == N.JITTER_EXPLICIT:
        want = (opts.y_dims, opts.x_dims, opts.antialias_samples, 2)
        if jitter is None or tuple(jitter.shape) != want:
            raise ValueError("explicit jitter must be [y_dims,x_dims,A,2] = %s" % (want,))
    if S > N.MAX_SHAPES:
        raise ValueError("at most %d shapes" % N.MAX_SHAPES)


class RenderFn(torch.autograd.Function):
    """image = RenderFn.apply(w2o, material, light, cam, kinds, jitter, opts)

    w2o      [S,3,4] | [B,S,3,4]  rows 0..2 of each shape's ``w2o.m``
    material [S,7]   | [B,S,7]    r,g,b,ks,kd,ka,shininess
    light    [6]                  lights[0]: direction, intensity
    cam      [16]    | [B,16]     camera o2w rows 0..2 + look_at (not differentiated)
    kinds    uint8 [S] or None    0 sphere / 1 square
    jitter   [y,x,A,2] or None    explicit sampleDist (opts.jitter_mode == EXPLICIT)
    ->       [rows,ncols,3] | [B,rows,ncols,3] float32
    """

    @staticmethod
    def forward(ctx, w2o, material, light, cam, kinds, jitter, opts):
        w2o = w2o.detach().contiguous().float()
        material = material.detach().contiguous().float()
        light = light.detach().contiguous().float()
        cam = cam.detach().contiguous().float()
        if jitter is not None:
            jitter = jitter.detach().contiguous().float()
        _check_inputs(w2o, material, light, cam, kinds, jitter, opts)
        L = N.lib()
        cull = opts.cull
        if cull and opts.jitter_mode == N.JITTER_EXPLICIT and not _jitter_allows_cull(jitter):
            cull = False
        sc, vw, S, B = _structs(w2o, material, light, cam, kinds, jitter, opts, 0, cull)
        batched = w2o.dim() == 4 or material.dim() == 3 or cam.dim() == 2
        dev = w2o.device
        nbytes = L.rrt_workspace_bytes(ctypes.byref(sc), ctypes.byref(vw))
        if nbytes == 0:
            raise RuntimeError("rrt_workspace_bytes rejected the scene/view")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        image = torch.empty((B, opts.rows, opts.ncols, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            N.check(L.rrt_render_fwd(ctypes.byref(sc), ctypes.byref(vw), _ptr(image), _ptr(ws), nbytes, stream),
                    "rrt_render_fwd")
        ctx.save_for_backward(w2o, material, light, cam)
        ctx.kinds, ctx.jitter, ctx.opts, ctx.ws, ctx.batched, ctx.cull = kinds, jitter, opts, ws, batched, cull
        return image if batched else image[0]

    @staticmethod
    def backward(ctx, grad_image):
        w2o, material, light, cam = ctx.saved_tensors
        opts = ctx.opts
        L = N.lib()
        g = grad_image.detach().contiguous().float()
        # the packed scene written by forward is still in ctx.ws
        sc, vw, S, B = _structs(w2o, material, light, cam, ctx.kinds, ctx.jitter, opts, N.FLAG_REUSE_SETUP, ctx.cull)
        dev = w2o.device
        flat = torch.empty(w2o.numel() + material.numel() + 6, dtype=torch.float32, device=dev)
        g_w2o = flat[:w2o.numel()].view(w2o.shape)
        g_mat = flat[w2o.numel():w2o.numel() + material.numel()].view(material.shape)
        g_light = flat[w2o.numel() + material.numel():]
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            N.check(L.rrt_render_bwd(ctypes.byref(sc), ctypes.byref(vw), _ptr(g), _ptr(g_w2o), _ptr(g_mat),
                                     _ptr(g_light), _ptr(ctx.ws), ctx.ws.numel(), stream), "rrt_render_bwd")
        hook = _grad_hook[0]
        if hook is not None:
            hook(flat)
        return g_w2o, g_mat, g_light, None, None, None, None


# Set by rrt.dist.sharded(): called with the flat [g_w2o | g_material | g_light]
# buffer right after the backward kernels (one all-reduce over the row bands).
_grad_hook = [None]


_default_cams = {}


def default_cam(device):
    """The root Camera's cam[16] (identity o2w, look_at (0,0,1)); one upload per device."""
    t = _default_cams.get(str(device))
    if t is None:
        t = torch.tensor(_IDENT_CAM, dtype=torch.float32, device=device)
        _default_cams[str(device)] = t
    return t


def render(w2o, material, light, kinds=None, cam=None, *, x_dims, y_dims, antialias_samples=4,
           shader="phong", specular=True, max_depth=1.0, transpose=None, jitter=None,
           row0=0, rows=None, cull=True, shadows=False):
    """Functional form of ``Scene(...).build(antialias_samples)``.

    jitter: None or "center" (U = 0.5), an int (hash seed: a reproducible
    stand-in for scene.py:24-25's unseeded draws) or a [y_dims,x_dims,A,2]
    tensor of sampleDist_x/_y (the reference's values (U + sample) / A lie in
    [0, 1]; a tensor with values outside that range is rendered without the
    tile cull, whose bounds assume it).  ``shadows=True`` is an extension the
    reference never renders (its ``Sphere.shadow`` is dormant, shape.py:85-97):
    a Phong-shaded point is black when another Sphere lies between it and the
    light; no gradient flows through the shadow mask.  ``transpose`` defaults to the root-camera
    semantics when ``cam`` is None and to the orbit semantics otherwise.
    """
    dev = w2o.device
    if w2o.shape[-2:] == (4, 4):
        w2o = w2o[..., :3, :]
    if cam is None:
        cam = default_cam(dev)
        if transpose is None:
            transpose = True
    elif transpose is None:
        transpose = False
    jit_t = None
    if jitter is None or (isinstance(jitter, str) and jitter == "center"):
        mode, seed = N.JITTER_CENTER, 0
    elif isinstance(jitter, (int, np.integer)):
        mode, seed = N.JITTER_HASH, int(jitter)
    else:
        mode, seed, jit_t = N.JITTER_EXPLICIT, 0, torch.as_tensor(jitter, dtype=torch.float32, device=dev)
    if kinds is not None and not torch.is_tensor(kinds):
        kinds = torch.as_tensor(np.asarray(kinds, dtype=np.uint8), device=dev)
    opts = RenderOptions(x_dims, y_dims, antialias_samples, shader, specular, max_depth, transpose,
                         mode, seed, row0, rows, cull, shadows)
    return RenderFn.apply(w2o, material, light, cam, kinds, jit_t, opts)
