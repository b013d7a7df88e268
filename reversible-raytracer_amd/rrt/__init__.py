"""rrt -- MI355X-native differentiable sphere raytracer (hot path of
lebek/reversible-raytracer behind the reference's own Scene/Shape/Shader API).

    from rrt.scene import *; from rrt.shader import *; from rrt.transform import *
    from rrt.optimize import GDOptimizer

mirror the reference's flat modules.  The per-pixel render and its gradient run
in hand-written gfx950 HIP kernels (csrc/, C ABI include/rrt.h); there is no
CPU render path.
"""
import os as _os

# kernel arguments in device memory (the HIP runtime's default on ROCm 7; host-resident kernargs cost every kernel of the
# render chain ~2.3 us: DESIGN.md section 7b).  Only effective if the runtime has not been initialised yet.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import config  # noqa: F401,E402
from .config import set_device, shared  # noqa: F401
from .optimize import GDOptimizer, MGDAutoOptimizer  # noqa: F401
from .render import RenderFn, RenderOptions, render  # noqa: F401
from .scene import Camera, Light, Material, Scene  # noqa: F401
from .shader import DepthMapShader, PhongShader  # noqa: F401
from .shape import Sphere, Square  # noqa: F401
from .transform import RayField, Transform, identity, rotate, scale, translate  # noqa: F401
from .util import broadcasted_switch, draw, drawWithMarkers, get_epsilon  # noqa: F401

__version__ = "0.1.0"
