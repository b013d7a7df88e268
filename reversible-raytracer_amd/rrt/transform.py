"""4x4 transform algebra -- same names and argument orders as the reference's
``transform.py`` (Transform :27-47, identity :56-58, translate :60-75,
scale :78-93, rotate :95-122, RayField :21-24), on torch tensors.

Tiny, O(1) work per shape: it stays ordinary (differentiable) torch code that
feeds the render kernels their ``w2o`` matrices.
"""
import math

import torch

from .config import as_tensor_variable, device, floatX


class RayField(object):
    """transform.py:21-24"""

    def __init__(self, origin, directions):
        self.origin = as_tensor_variable(origin)
        self.rays = as_tensor_variable(directions)


class Transform(object):
    """transform.py:27-47: ``m`` and ``mInv`` kept in lock-step."""

    def __init__(self, m, mInv):
        self.m = m
        self.mInv = mInv

    def inverse(self):
        return Transform(self.mInv, self.m)

    def __mul__(self, other):
        m = self.m @ other.m
        mInv = other.mInv @ self.mInv
        return Transform(m, mInv)

    def __call__(self, x):
        """Apply to a RayField (transform.py:40-47).  NOTE the reference's
        ``.T`` at :46 reverses all axes of the [4,P,Q] product, so the returned
        ray grid is the spatial transpose [Q,P,3] of the input grid -- kept."""
        if isinstance(x, RayField):
            o = x.origin
            r = x.rays
            one = torch.ones(1, dtype=o.dtype, device=o.device)
            origin = (self.m @ torch.cat([o[:3], one]))[:3]
            rays = torch.cat([r, torch.zeros_like(r)[:, :, :1]], dim=2)
            rays = torch.tensordot(self.m, rays, dims=([1], [2])).permute(2, 1, 0)[:, :, :3]
            return RayField(origin, rays)
        raise TypeError("Transform can only be applied to a RayField")


def identity():
    """transform.py:56-58"""
    return Transform(_eye(device()).clone(), _eye(device()).clone())


_consts = {}


def _const(name, dev, make):
    """Constant helper tensors, built once per device (on the host, one upload):
    element writes like ``t[3, 3] = 1`` are host-to-device copies, which a HIP
    graph capture of the step does not allow."""
    key = (name, str(dev))
    t = _consts.get(key)
    if t is None:
        t = make().to(dev)
        _consts[key] = t
    return t


def _basis(positions, dev):
    def make():
        b = torch.zeros(len(positions), 4, 4, dtype=floatX)
        for n, (r, c) in enumerate(positions):
            b[n, r, c] = 1.0
        return b
    return _const(("basis",) + tuple(positions), dev, make)


def _last(dev):
    def make():
        t = torch.zeros(4, 4, dtype=floatX)
        t[3, 3] = 1.0
        return t
    return _const("last", dev, make)


def _eye(dev):
    return _const("eye", dev, lambda: torch.eye(4, dtype=floatX))


def _place(vals, positions):
    """4x4 matrix with vals[n] at positions[n] (functional set_subtensor chain)."""
    dev = vals[0].device
    return (torch.stack(list(vals)).view(-1, 1, 1) * _basis(tuple(positions), dev)).sum(0)


def translate(x):
    """transform.py:60-75"""
    x = as_tensor_variable(x)
    eye = _eye(x.device)
    pos = [(0, 3), (1, 3), (2, 3)]
    off = _place([x[0], x[1], x[2]], pos)
    return Transform(eye + off, eye - off)


def scale(x):
    """transform.py:78-93 (inverse uses 1./x)"""
    x = as_tensor_variable(x)
    last = _last(x.device)
    pos = [(0, 0), (1, 1), (2, 2)]
    m = _place([x[0], x[1], x[2]], pos) + last
    mInv = _place([1.0 / x[0], 1.0 / x[1], 1.0 / x[2]], pos) + last
    return Transform(m, mInv)


def rotate(angle, axis):
    """transform.py:95-122: angle in DEGREES about a (unit) axis; inverse = m.T"""
    angle = as_tensor_variable(angle)
    a = as_tensor_variable(axis)
    radians = angle * math.pi / 180.0
    s = torch.sin(radians)
    c = torch.cos(radians)
    vals = [
        a[0] * a[0] + (1.0 - a[0] * a[0]) * c,
        a[0] * a[1] * (1.0 - c) - a[2] * s,
        a[0] * a[2] * (1.0 - c) + a[1] * s,
        a[0] * a[1] * (1.0 - c) + a[2] * s,
        a[1] * a[1] + (1.0 - a[1] * a[1]) * c,
        a[1] * a[2] * (1.0 - c) - a[0] * s,
        a[0] * a[2] * (1.0 - c) - a[1] * s,
        a[1] * a[2] * (1.0 - c) + a[0] * s,
        a[2] * a[2] + (1.0 - a[2] * a[2]) * c,
    ]
    pos = [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2), (2, 0), (2, 1), (2, 2)]
    m = _place(vals, pos) + _last(angle.device)
    return Transform(m, m.T)
