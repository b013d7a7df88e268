"""Shapes -- reference ``shape.py`` (Square :16-69, Sphere :72-138).

``Scene.build()`` never calls ``distance`` / ``normals``: the HIP kernels
intersect and shade in registers.  The two methods exist because they are part
of the reference's public surface (explicit ``RayField`` in, per-ray tensors
out); they are plain torch and follow the cited lines, quirks included.
"""
import torch

from .transform import RayField  # noqa: F401  (re-exported like the reference's `from transform import *`)

KIND_SPHERE = 0
KIND_SQUARE = 1


class Shape(object):
    kind = KIND_SPHERE

    def __init__(self, name=None):
        pass

    def setTransform(self, o2w):
        self.o2w = o2w
        self.w2o = o2w.inverse()


class Square(Shape):
    """Unit square on the object xy-plane, normal (0,0,1) (shape.py:16-23)."""
    kind = KIND_SQUARE

    def __init__(self, o2w, material):
        self.o2w = o2w
        self.w2o = o2w.inverse()
        self.material = material

    def _hit(self, rays, origin):
        """shape.py:25-40"""
        rz = rays[:, :, 2]
        not_parallel = rz != 0
        ts = -origin[2] / torch.where(not_parallel, rz, torch.ones_like(rz))
        inter = origin + ts.unsqueeze(-1) * rays
        mask = (inter[:, :, 0] > -0.5) & (inter[:, :, 0] < 0.5) & (inter[:, :, 1] > -0.5) & \
               (inter[:, :, 1] < 0.5) & (ts > 0) & not_parallel
        ts = torch.where(mask, ts, torch.full_like(ts, float("inf")))
        return mask, ts

    def distance(self, rayField):
        """shape.py:43-50"""
        rf = self.w2o(rayField)
        return self._hit(rf.rays, rf.origin)[1]

    def normals(self, rayField):
        """shape.py:52-69"""
        rf = self.w2o(rayField)
        mask, _ = self._hit(rf.rays, rf.origin)
        n = torch.zeros_like(rf.rays)
        n[:, :, 2] = torch.where(rf.origin[2] > 0, 1.0, -1.0)
        return n * mask.unsqueeze(-1).to(n.dtype)


class Sphere(Shape):
    """Unit sphere in object space (shape.py:72-76)."""
    kind = KIND_SPHERE

    def __init__(self, o2w, material):
        self.o2w = o2w
        self.w2o = o2w.inverse()
        self.material = material

    def _hit(self, rays, origin):
        """shape.py:78-83 (discriminant)"""
        pnorm = torch.dot(origin, origin)
        vnorm = torch.sum(rays * rays, dim=2)
        pdotv = torch.tensordot(rays, origin, dims=1)
        return pdotv * pdotv - vnorm * (pnorm - 1)

    def distance(self, rayField):
        """shape.py:109-126: +inf on miss, no t > 0 test."""
        rf = self.w2o(rayField)
        pdotv = torch.tensordot(rf.rays, rf.origin, dims=1)
        vnorm = torch.sum(rf.rays * rf.rays, dim=2)
        det = self._hit(rf.rays, rf.origin)
        bad = (det <= 0) | torch.isnan(det)
        root = torch.sqrt(torch.where(bad, torch.ones_like(det), det))
        d = torch.minimum((-pdotv - root) / vnorm, (-pdotv + root) / vnorm)
        return torch.where(bad, torch.full_like(d, float("inf")), d)

    def normals(self, rayField):
        """shape.py:128-138: object-space normals ('need to fix' in the reference)."""
        rf = self.w2o(rayField)
        distance = self.distance(rayField)
        distance = torch.where(torch.isinf(distance), torch.zeros_like(distance), distance)
        proj = rf.origin + distance.unsqueeze(-1) * rf.rays
        return proj / torch.sqrt(torch.sum(proj ** 2, dim=2)).unsqueeze(-1)
