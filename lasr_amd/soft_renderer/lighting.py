"""Lighting stage (reference API: soft_renderer/lighting.py:9-67 under /root/reference/third_party/softras/)."""
import torch
import torch.nn as nn

from . import functional as srf


class AmbientLighting(nn.Module):
    def __init__(self, light_intensity=0.5, light_color=(1, 1, 1)):
        super().__init__()
        self.light_intensity, self.light_color = light_intensity, light_color

    def forward(self, light):
        return srf.ambient_lighting(light, self.light_intensity, self.light_color)


class DirectionalLighting(nn.Module):
    def __init__(self, light_intensity=0.5, light_color=(1, 1, 1), light_direction=(0, 1, 0)):
        super().__init__()
        self.light_intensity, self.light_color, self.light_direction = light_intensity, light_color, light_direction

    def forward(self, light, normals):
        return srf.directional_lighting(light, normals, self.light_intensity, self.light_color, self.light_direction)


class Lighting(nn.Module):
    def __init__(self, light_mode='surface', intensity_ambient=0.5, color_ambient=[1, 1, 1],
                 intensity_directionals=0.5, color_directionals=[1, 1, 1], directions=[0, 1, 0]):
        super().__init__()
        if light_mode not in ['surface', 'vertex']:
            raise ValueError('Lighting mode only support surface and vertex')
        self.light_mode = light_mode
        self.ambient = AmbientLighting(intensity_ambient, color_ambient)
        self.directionals = nn.ModuleList([DirectionalLighting(intensity_directionals, color_directionals,
                                                               directions)])

    def _constant_light(self):
        """[r,g,b] when no directional light is on and the ambient colour is a python constant, else None."""
        if any(d.light_intensity != 0 for d in self.directionals):
            return None
        c = self.ambient.light_color
        if not isinstance(c, (list, tuple)) or not isinstance(self.ambient.light_intensity, (int, float)):
            return None
        return [float(self.ambient.light_intensity * v) for v in c]

    def forward(self, mesh):
        per_vertex = self.light_mode == 'vertex'
        const = self._constant_light()
        if const is not None:                      # light = 0 + intensity * colour, the same for every vertex / face
            if any(v != 1.0 for v in const):       # (x * 1 is x: LASR's ambient-only white light launches nothing)
                reps = mesh.textures.shape[-1] // 3
                mesh.textures = mesh.textures * srf.const_tensor(const * reps, mesh.device)
            return mesh
        shape_like = mesh.vertices if per_vertex else mesh.faces
        light = torch.zeros(shape_like.shape, dtype=torch.float32, device=mesh.device)
        light = self.ambient(light)
        for d in self.directionals:
            if d.light_intensity == 0:
                continue
            light = d(light, mesh.vertex_normals if per_vertex else mesh.surface_normals)
        if per_vertex and mesh.textures.shape[-1] != 3:        # [B,V,3k] attribute stacks: the same light per triple
            light = light.repeat(1, 1, mesh.textures.shape[-1] // 3)
        mesh.textures = mesh.textures * (light if per_vertex else light[:, :, None, :])
        return mesh
