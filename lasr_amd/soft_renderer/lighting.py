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

    def forward(self, mesh):
        per_vertex = self.light_mode == 'vertex'
        shape_like = mesh.vertices if per_vertex else mesh.faces
        light = torch.zeros(shape_like.shape, dtype=torch.float32, device=mesh.device)
        light = self.ambient(light)
        for d in self.directionals:
            # LASR renders with intensity_directionals = 0 (nnutils/mesh_net.py:136-149): the reference still
            # builds the normals (3 index_add_ per call) only to multiply them by zero; skip that work.
            if d.light_intensity == 0:
                continue
            light = d(light, mesh.vertex_normals if per_vertex else mesh.surface_normals)
        if per_vertex and mesh.textures.shape[-1] != 3:        # [B,V,3k] attribute stacks: the same light per triple
            light = light.repeat(1, 1, mesh.textures.shape[-1] // 3)
        mesh.textures = mesh.textures * (light if per_vertex else light[:, :, None, :])
        return mesh
