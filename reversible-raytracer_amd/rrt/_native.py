"""ctypes binding of ``librrt_hip.so`` (C ABI: ``include/rrt.h``).

The library is built in-tree by :func:`build_native` (``hipcc
--offload-arch=gfx950``) and loaded from this directory.  There is no
fallback: if the shared object is missing or a call fails, a RuntimeError is
raised -- the product path never silently renders on the CPU.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
_REPO = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
INCLUDE = os.path.join(_REPO, "include")
LIB_PATH = os.environ.get("RRT_LIB", os.path.join(_HERE, "librrt_hip.so"))   # RRT_LIB: diagnostic builds only

RRT_VERSION = 100
MAX_SHAPES = 1024

KIND_SPHERE, KIND_SQUARE = 0, 1
SHADER_PHONG, SHADER_DEPTH = 0, 1
JITTER_CENTER, JITTER_EXPLICIT, JITTER_HASH = 0, 1, 2
FLAG_NO_CULL, FLAG_REUSE_SETUP, FLAG_SKIP_FINALIZE, FLAG_SHADOWS = 1, 2, 4, 8

# every symbol include/rrt.h declares
EXPORTS = ("rrt_version", "rrt_strerror", "rrt_last_hip_error", "rrt_image_cols",
           "rrt_image_rows_full", "rrt_workspace_bytes", "rrt_render_fwd", "rrt_render_bwd")


class RrtScene(ctypes.Structure):
    _fields_ = [("S", ctypes.c_int32), ("B", ctypes.c_int32),
                ("w2o", ctypes.c_void_p), ("w2o_bstride", ctypes.c_int64),
                ("material", ctypes.c_void_p), ("material_bstride", ctypes.c_int64),
                ("light", ctypes.c_void_p), ("kind", ctypes.c_void_p)]


class RrtView(ctypes.Structure):
    _fields_ = [("x_dims", ctypes.c_int32), ("y_dims", ctypes.c_int32),
                ("antialias_samples", ctypes.c_int32),
                ("row0", ctypes.c_int32), ("rows", ctypes.c_int32),
                ("shader_id", ctypes.c_int32), ("specular", ctypes.c_int32),
                ("transpose", ctypes.c_int32), ("jitter_mode", ctypes.c_int32),
                ("seed", ctypes.c_uint32), ("flags", ctypes.c_uint32),
                ("max_depth", ctypes.c_float),
                ("cam", ctypes.c_void_p), ("cam_bstride", ctypes.c_int64),
                ("jitter", ctypes.c_void_p)]


def sources():
    return [os.path.join(CSRC, "rrt_kernels.hip")]


def build_native(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into rrt/librrt_hip.so (cross-compiles
    without a GPU).  Returns the path."""
    srcs = sources()
    deps = srcs + [os.path.join(CSRC, "rrt_device.h"), os.path.join(INCLUDE, "rrt.h")]
    if (not force and os.path.exists(LIB_PATH)
            and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in deps)):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    # -fno-slp-vectorize: SLP-packed v_pk_*_f32 cost 8 issue cycles (= two scalar
    # ops) plus the v_mov shuffles that feed them -- a net loss in these kernels
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-shared",
           "-I", INCLUDE, "-I", CSRC] + srcs + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "librrt_hip.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'`; there is no CPU fallback for the render path" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    L.rrt_version.restype = ctypes.c_int
    L.rrt_strerror.restype = ctypes.c_char_p
    L.rrt_strerror.argtypes = [ctypes.c_int]
    L.rrt_last_hip_error.restype = ctypes.c_int
    L.rrt_image_cols.argtypes = [ctypes.POINTER(RrtView)]
    L.rrt_image_rows_full.argtypes = [ctypes.POINTER(RrtView)]
    L.rrt_workspace_bytes.restype = ctypes.c_size_t
    L.rrt_workspace_bytes.argtypes = [ctypes.POINTER(RrtScene), ctypes.POINTER(RrtView)]
    L.rrt_render_fwd.restype = ctypes.c_int
    L.rrt_render_fwd.argtypes = [ctypes.POINTER(RrtScene), ctypes.POINTER(RrtView), ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.rrt_render_bwd.restype = ctypes.c_int
    L.rrt_render_bwd.argtypes = [ctypes.POINTER(RrtScene), ctypes.POINTER(RrtView), ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    v = L.rrt_version()
    if v != RRT_VERSION:
        raise RuntimeError("librrt_hip.so version %d != binding version %d" % (v, RRT_VERSION))
    _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        L = lib()
        msg = L.rrt_strerror(rc).decode()
        raise RuntimeError("%s failed: %s (status %d, hip error %d)" % (what, msg, rc, L.rrt_last_hip_error()))
