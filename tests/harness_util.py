"""ctypes loader for tests/cpu_harness (the product's device math compiled for
the host).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, "cpu_harness", "harness.cpp")
HDR = os.path.join(REPO, "reversible-raytracer_amd", "csrc", "rrt_device.h")
OUT = os.path.join(HERE, "cpu_harness", "_build", "libharness.so")


class HView(ctypes.Structure):
    _fields_ = [("x_dims", ctypes.c_int32), ("y_dims", ctypes.c_int32), ("A", ctypes.c_int32),
                ("row0", ctypes.c_int32), ("rows", ctypes.c_int32), ("shader", ctypes.c_int32),
                ("specular", ctypes.c_int32), ("transpose", ctypes.c_int32),
                ("jitter_mode", ctypes.c_int32), ("seed", ctypes.c_uint32),
                ("max_depth", ctypes.c_float), ("force_generic", ctypes.c_int32),
                ("shadows", ctypes.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    need = (not os.path.exists(OUT)) or os.path.getmtime(OUT) < max(os.path.getmtime(SRC), os.path.getmtime(HDR))
    if need:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-ffp-contract=off",
                               "-I", os.path.dirname(HDR), SRC, "-o", OUT])
    _lib = ctypes.CDLL(OUT)
    _lib.harness_cull_check.restype = ctypes.c_int64
    return _lib


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a, ty=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(ty)) if a is not None else None


IDENT_CAM = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0], dtype=np.float32)


def make_cam(cam_o2w=None, look_at=(0, 0, 1)):
    cam = IDENT_CAM.copy()
    if cam_o2w is not None:
        cam[:12] = np.asarray(cam_o2w, dtype=np.float64)[:3, :4].reshape(-1)
    cam[12:15] = look_at
    return cam


def _view(x_dims, y_dims, A, transpose, shader, specular, max_depth, jitter, seed, row0, rows, force_generic=False,
          shadows=False):
    n_full = x_dims if transpose else y_dims
    if rows is None:
        rows = n_full - row0
    if jitter is None:
        mode = 0
    elif isinstance(jitter, (int, np.integer)):
        mode, seed, jitter = 2, int(jitter), None
    else:
        mode = 1
    hv = HView(x_dims, y_dims, A, row0, rows, 0 if shader == "phong" else 1, int(bool(specular)),
               int(bool(transpose)), mode, seed, float(max_depth), int(bool(force_generic)), int(bool(shadows)))
    jit = _f32(jitter) if mode == 1 else None
    ncols = y_dims if transpose else x_dims
    return hv, jit, rows, ncols


def fwd(w2o, material, light, kinds=None, *, x_dims, y_dims, A=1, cam=None, transpose=True,
        shader="phong", specular=True, max_depth=1.0, jitter=None, seed=0, row0=0, rows=None,
        double=True, want_ids=False, force_generic=False, shadows=False):
    """jitter: None (centre), int (hash seed) or [y,x,A,2] sampleDist array."""
    w2o = _f32(np.asarray(w2o)[:, :3, :]); material = _f32(material); light = _f32(light)
    S = w2o.shape[0]
    kind = np.ascontiguousarray(np.asarray(kinds if kinds is not None else [0] * S, dtype=np.uint8))
    cam = _f32(cam if cam is not None else IDENT_CAM)
    hv, jit, rows, ncols = _view(x_dims, y_dims, A, transpose, shader, specular, max_depth, jitter, seed, row0, rows,
                                 force_generic, shadows)
    image = np.zeros((rows, ncols, 3), dtype=np.float64)
    ids = np.zeros((A, rows, ncols), dtype=np.int32) if want_ids else None
    lib().harness_fwd(int(double), S, _ptr(w2o), _ptr(material), _ptr(light), _ptr(kind, ctypes.c_uint8),
                      _ptr(cam), ctypes.byref(hv), _ptr(jit), _ptr(image, ctypes.c_double),
                      _ptr(ids, ctypes.c_int32) if want_ids else None)
    return (image, ids) if want_ids else image


def bwd(w2o, material, light, kinds=None, *, grad_image, x_dims, y_dims, A=1, cam=None, transpose=True,
        shader="phong", specular=True, max_depth=1.0, jitter=None, seed=0, row0=0, rows=None, double=True,
        force_generic=False, shadows=False):
    w2o = _f32(np.asarray(w2o)[:, :3, :]); material = _f32(material); light = _f32(light)
    S = w2o.shape[0]
    kind = np.ascontiguousarray(np.asarray(kinds if kinds is not None else [0] * S, dtype=np.uint8))
    cam = _f32(cam if cam is not None else IDENT_CAM)
    hv, jit, rows, ncols = _view(x_dims, y_dims, A, transpose, shader, specular, max_depth, jitter, seed, row0, rows,
                                 force_generic, shadows)
    g = np.ascontiguousarray(np.asarray(grad_image, dtype=np.float64))
    assert g.shape == (rows, ncols, 3)
    gw = np.zeros((S, 3, 4)); gm = np.zeros((S, 7)); gl = np.zeros(6)
    lib().harness_bwd(int(double), S, _ptr(w2o), _ptr(material), _ptr(light), _ptr(kind, ctypes.c_uint8),
                      _ptr(cam), ctypes.byref(hv), _ptr(jit), _ptr(g, ctypes.c_double),
                      _ptr(gw, ctypes.c_double), _ptr(gm, ctypes.c_double), _ptr(gl, ctypes.c_double))
    return gw, gm, gl


def cull_check(w2o, material, light, kinds=None, *, x_dims, y_dims, A=1, cam=None, transpose=True,
               jitter=None, seed=0, row0=0, rows=None, tw=64, th=16, want_hist=False, check_rays=True):
    w2o = _f32(np.asarray(w2o)[:, :3, :]); material = _f32(material); light = _f32(light)
    S = w2o.shape[0]
    kind = np.ascontiguousarray(np.asarray(kinds if kinds is not None else [0] * S, dtype=np.uint8))
    cam = _f32(cam if cam is not None else IDENT_CAM)
    hv, jit, rows, ncols = _view(x_dims, y_dims, A, transpose, "phong", True, 1.0, jitter, seed, row0, rows)
    mean = ctypes.c_double(0)
    hist = np.zeros(4, dtype=np.int64)
    v = lib().harness_cull_check(S, _ptr(w2o), _ptr(material), _ptr(light), _ptr(kind, ctypes.c_uint8),
                                 _ptr(cam), ctypes.byref(hv), _ptr(jit), tw, th, ctypes.byref(mean),
                                 _ptr(hist, ctypes.c_int64), int(bool(check_rays)))
    return (int(v), mean.value, hist) if want_hist else (int(v), mean.value)
