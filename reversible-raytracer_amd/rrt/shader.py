"""Shaders -- reference ``shader.py`` (DepthMapShader :9-20, PhongShader
:23-53; the orbit fork drops the specular term, orbit_experiments/shader.py:45,48).

For ``Scene.build()`` a shader is only a tag (+ maxDepth / specular flag) that
selects the kernel's shading branch.  ``shade`` is kept for API parity with the
reference (explicit per-ray tensors, plain torch) and is not on the hot path.
"""
import torch

from .config import as_tensor_variable
from .util import broadcasted_switch


class Shader(object):
    shader_id = "phong"
    specular = True
    maxDepth = 1.0


class DepthMapShader(Shader):
    shader_id = "depth"

    def __init__(self, maxDepth):
        self.maxDepth = float(maxDepth)

    def shade(self, shape, lights, camera):
        """shader.py:14-20: 1 - t/maxDepth on three channels; no clip, no inf mask."""
        distance = shape.distance(camera.rays)
        scaled = distance / self.maxDepth
        return (1 - scaled).unsqueeze(-1) * torch.ones(3, dtype=distance.dtype, device=distance.device)


class PhongShader(Shader):
    shader_id = "phong"

    def __init__(self, specular=True):
        # specular=False reproduces orbit_experiments/shader.py (ks term commented out)
        self.specular = bool(specular)

    def shade(self, shape, lights, camera):
        """shader.py:28-53"""
        light = lights[0]
        material = shape.material
        normals = shape.normals(camera.rays)
        L = light.normed_dir()
        ndotl = torch.tensordot(normals, -L, dims=1)
        phong = material.ka + material.kd * ndotl
        if self.specular:
            rm = 2.0 * ndotl.unsqueeze(-1) * normals + L
            look = as_tensor_variable(camera.look_at)
            phong = phong + material.ks * torch.pow(torch.tensordot(rm, look, dims=1), material.shininess)
        colorized = phong.unsqueeze(-1) * material.color.view(1, 1, 3) * light.intensity.view(1, 1, 3)
        clipped = torch.clamp(colorized, 0, 1)
        distances = shape.distance(camera.rays)
        return broadcasted_switch(torch.isinf(distances), torch.zeros_like(clipped), clipped)
