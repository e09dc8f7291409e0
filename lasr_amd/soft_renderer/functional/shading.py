"""Ambient / directional light accumulation (reference: soft_renderer/functional/ambient_lighting.py:7-19,
directional_lighting.py:7-29 under /root/reference/third_party/softras/)."""
import numpy as np
import torch
import torch.nn.functional as F


def _rgb(x, device):
    if isinstance(x, (tuple, list)):
        from .cameras import const_tensor
        x = const_tensor(x, device)
    elif isinstance(x, np.ndarray):
        x = torch.from_numpy(x).float().to(device)
    return x[None, :] if x.ndimension() == 1 else x


def ambient_lighting(light, light_intensity=0.5, light_color=(1, 1, 1)):
    light += light_intensity * _rgb(light_color, light.device)[:, None, :]
    return light


def directional_lighting(light, normals, light_intensity=0.5, light_color=(1, 1, 1), light_direction=(0, 1, 0)):
    color = _rgb(light_color, light.device)
    direction = _rgb(light_direction, light.device)
    cosine = F.relu(torch.sum(normals * direction, dim=2))
    light += light_intensity * (color[:, None, :] * cosine[:, :, None])
    return light
