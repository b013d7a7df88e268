"""Scene / Camera / Light / Material -- reference ``scene.py`` (Scene :11-52,
Camera :55-75, Light :78-86, Material :89-101) and the orbit fork's camera
(``orbit_experiments/scene.py:55-80``).

``Scene.build()`` is where the reference unrolls its A x S Theano graph; here
it packs the scene into four small tensors and calls the HIP render operator
(``rrt.render.RenderFn``): ray generation, intersection, shading, composite
and the anti-alias average all happen inside one kernel, and autograd reaches
the shapes' transforms / materials / the light through its analytic adjoint.
"""
import numpy as np
import torch

from . import _native as N
from .config import as_tensor_variable, device, floatX
from .render import RenderFn, RenderOptions
from .shape import *       # noqa: F401,F403  (the reference's scene.py does `from shape import *`)
from .transform import *   # noqa: F401,F403
from .transform import RayField, Transform
from .util import *        # noqa: F401,F403


_kinds_cache = {}


class Scene(object):
    def __init__(self, shapes, lights, camera, shader):
        self.shapes = shapes
        self.lights = lights
        self.camera = camera
        self.shader = shader
        self._seed = None

    # -- packing: what the kernels read (include/rrt.h) ----------------------
    def pack(self):
        """(w2o[S,3,4], material[S,7], light[6], kinds uint8[S], cam[16])."""
        dev = device()
        S = len(self.shapes)
        if S:
            w2o = torch.stack([s.w2o.m[:3, :] for s in self.shapes])
            material = torch.stack([s.material.as_row() for s in self.shapes])
            key = (tuple(s.kind for s in self.shapes), str(dev))
            kinds = _kinds_cache.get(key)          # module-level: drivers rebuild their Scene every step
            if kinds is None:
                kinds = torch.as_tensor(np.array(key[0], dtype=np.uint8), device=dev)
                _kinds_cache[key] = kinds
        else:
            w2o = torch.zeros(0, 3, 4, dtype=floatX, device=dev)
            material = torch.zeros(0, 7, dtype=floatX, device=dev)
            kinds = None
        light = self.lights[0].as_row()        # only lights[0] is used (shader.py:33)
        return w2o, material, light, kinds, self.camera.as_row()

    def build(self, antialias_samples=4, jitter=None, row0=0, rows=None, shadows=False):
        """scene.py:18-52 -> image tensor (x_dims, y_dims, 3), attached to autograd.

        jitter: None -> the reference draws unseeded uniform jitter once, at
        graph-build time, and bakes it into the graph as constants
        (scene.py:24-25); here a hash seed is drawn once per Scene (from
        ``np.random``, like the reference) and reused by every later build(),
        so repeated calls see the same 'baked' jitter.  Pass "center", an int
        seed, or an explicit [y_dims,x_dims,A,2] sampleDist tensor to control it
        (values in [0, 1] like the reference's (U + sample) / A; anything else is
        rendered with the tile cull off, see rrt.render).
        ``row0/rows`` render a band of image rows (multi-GPU sharding).
        ``shadows=True``: hard shadows after the dormant ``Sphere.shadow`` (shape.py:85-97; the reference's
        own call site, scene.py:41-45, is commented out) -- an extension, off by default.
        """
        cam = self.camera
        jit_t = None
        if jitter is None:
            if self._seed is None:
                self._seed = int(np.random.randint(0, 2 ** 31 - 1))
            mode, seed = N.JITTER_HASH, self._seed
        elif isinstance(jitter, str):
            if jitter != "center":
                raise ValueError("jitter must be None, 'center', an int seed or a tensor")
            mode, seed = N.JITTER_CENTER, 0
        elif isinstance(jitter, (int, np.integer)):
            mode, seed = N.JITTER_HASH, int(jitter)
        else:
            mode, seed = N.JITTER_EXPLICIT, 0
            jit_t = torch.as_tensor(jitter, dtype=floatX, device=device())
        opts = RenderOptions(cam.x_dims, cam.y_dims, antialias_samples, self.shader.shader_id,
                             getattr(self.shader, "specular", True), getattr(self.shader, "maxDepth", 1.0),
                             cam.transpose, mode, seed, row0, rows, True, shadows)
        w2o, material, light, kinds, cam_row = self.pack()
        return RenderFn.apply(w2o, material, light, cam_row, kinds, jit_t, opts)


class Camera(object):
    """Root: ``Camera(x_dims, y_dims)`` (scene.py:55-59; origin 0, look_at
    (0,0,1)).  Orbit: ``Camera(x_dims, y_dims, o2w, camera_dir)``
    (orbit_experiments/scene.py:55-61): rays are ``o2w(RayField(0, rays))``."""

    def __init__(self, x_dims, y_dims, o2w=None, camera_dir=None):
        self.x_dims = x_dims
        self.y_dims = y_dims
        self.o2w = o2w
        self.w2o = o2w.inverse() if o2w is not None else None
        self.look_at = np.asarray([0, 0, 1.0], dtype="float32") if camera_dir is None \
            else np.asarray(camera_dir, dtype="float32")
        self.rays = None
        self._row = None

    @property
    def transpose(self):
        """The root camera's rays go through ONE Transform.__call__ (the
        shape's w2o), each of which spatially transposes the grid
        (transform.py:46): image[a,b] shows ray[b,a].  The orbit camera adds a
        second one (its own o2w), and the two cancel."""
        return self.o2w is None

    def as_row(self):
        """cam[16] of include/rrt.h: rows 0..2 of o2w.m, look_at, pad."""
        dev = device()
        if self._row is not None and self._row[0] == str(dev):
            return self._row[1]          # cameras are not differentiated: built once (also keeps H2D copies out of graphs)
        if self.o2w is None:
            m = torch.eye(4, dtype=floatX, device=dev)[:3, :]
        else:
            m = self.o2w.m[:3, :].detach().to(dev)
        look = torch.as_tensor(self.look_at, dtype=floatX, device=dev)
        row = torch.cat([m.reshape(-1), look, torch.zeros(1, dtype=floatX, device=dev)])
        self._row = (str(dev), row)
        return row

    def make_rays(self, x_dims, y_dims, sampleDist_x=None, sampleDist_y=None):
        """scene.py:61-75 (NumPy, float64 then cast) -> RayField.  API parity
        only; the kernels generate rays in registers."""
        rays = np.dstack(np.meshgrid(np.linspace(0.5, -0.5, y_dims), np.linspace(-0.5, 0.5, x_dims),
                                     indexing="ij"))
        rays = np.dstack([rays, np.ones([y_dims, x_dims], dtype="float32")])
        rays = np.divide(rays, np.linalg.norm(rays, axis=2).reshape(y_dims, x_dims, 1).repeat(3, 2))
        rays = np.asarray(rays, dtype="float32")
        if sampleDist_x is not None:
            rays[:, :, 0] = rays[:, :, 0] + np.asarray(sampleDist_x, dtype="float32") / x_dims
        if sampleDist_y is not None:
            rays[:, :, 1] = rays[:, :, 1] + np.asarray(sampleDist_y, dtype="float32") / y_dims
        rf = RayField([0.0, 0.0, 0.0], rays)
        return self.o2w(rf) if self.o2w is not None else rf


class Light(object):
    def __init__(self, direction, intensity):
        self.direction = as_tensor_variable(direction)
        self.intensity = as_tensor_variable(intensity)

    def normed_dir(self):
        """scene.py:83-86"""
        d = self.direction
        norm = torch.sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2)
        return d / norm

    def as_row(self):
        return torch.cat([self.direction.reshape(3), self.intensity.reshape(3)])


class Material(object):
    def __init__(self, color, ks, kd, ka, shininess):
        """
        ks -- specular reflection constant
        kd -- diffuse reflection constant
        ka -- ambient reflection constant
        shininess -- shininess constant
        """
        self.ks = as_tensor_variable(ks)
        self.kd = as_tensor_variable(kd)
        self.ka = as_tensor_variable(ka)
        self.color = as_tensor_variable(color)
        self.shininess = as_tensor_variable(shininess)

    def as_row(self):
        """material[7] of include/rrt.h: r,g,b,ks,kd,ka,shininess."""
        return torch.cat([self.color.reshape(3), self.ks.reshape(1), self.kd.reshape(1), self.ka.reshape(1),
                          self.shininess.reshape(1)])
