"""Camera stage (reference API: soft_renderer/transform.py:9-110 under /root/reference/third_party/softras/)."""
import math

import numpy as np
import torch
import torch.nn as nn

from . import functional as srf


def _default_eye(viewing_angle):
    return [0, 0, -(1. / math.tan(math.radians(viewing_angle)) + 1)]


class Projection(nn.Module):
    def __init__(self, P, dist_coeffs=None, orig_size=512):
        super().__init__()
        if isinstance(P, np.ndarray):
            P = torch.from_numpy(P)
            P = P.cuda() if torch.cuda.is_available() else P
        if P is None or P.ndimension() != 3 or P.shape[1] != 3 or P.shape[2] != 4:
            raise ValueError('You need to provide a valid (batch_size)x3x4 projection matrix')
        if dist_coeffs is None:
            dist_coeffs = torch.zeros(P.shape[0], 5, dtype=torch.float32, device=P.device)
        self.P, self.dist_coeffs, self.orig_size = P, dist_coeffs, orig_size

    def forward(self, vertices):
        return srf.projection(vertices, self.P, self.dist_coeffs, self.orig_size)


class _EyeCamera(nn.Module):
    def __init__(self, perspective, viewing_angle, viewing_scale, eye):
        super().__init__()
        self.perspective, self.viewing_angle, self.viewing_scale = perspective, viewing_angle, viewing_scale
        self._eye = _default_eye(viewing_angle) if eye is None else eye

    def _project(self, vertices):
        if self.perspective:
            return srf.perspective(vertices, angle=self.viewing_angle)
        return srf.orthogonal(vertices, scale=self.viewing_scale)


class LookAt(_EyeCamera):
    def __init__(self, perspective=True, viewing_angle=30, viewing_scale=1.0, eye=None):
        super().__init__(perspective, viewing_angle, viewing_scale, eye)

    def forward(self, vertices):
        return self._project(srf.look_at(vertices, self._eye))


class Look(_EyeCamera):
    def __init__(self, camera_direction=[0, 0, 1], perspective=True, viewing_angle=30, viewing_scale=1.0, eye=None):
        super().__init__(perspective, viewing_angle, viewing_scale, eye)
        self.camera_direction = camera_direction

    def forward(self, vertices):
        return self._project(srf.look(vertices, self._eye, self.camera_direction))


class Transform(nn.Module):
    def __init__(self, camera_mode='projection', P=None, dist_coeffs=None, orig_size=512,
                 perspective=True, viewing_angle=30, viewing_scale=1.0, eye=None, camera_direction=[0, 0, 1]):
        super().__init__()
        self.camera_mode = camera_mode
        if camera_mode == 'projection':
            self.transformer = Projection(P, dist_coeffs, orig_size)
        elif camera_mode == 'look':
            self.transformer = Look(camera_direction, perspective, viewing_angle, viewing_scale, eye)
        elif camera_mode == 'look_at':
            self.transformer = LookAt(perspective, viewing_angle, viewing_scale, eye)
        else:
            raise ValueError('Camera mode has to be one of projection, look or look_at')

    def forward(self, mesh):
        mesh.vertices = self.transformer(mesh.vertices)
        return mesh

    def _need_eye(self):
        if self.camera_mode not in ['look', 'look_at']:
            raise ValueError('Projection does not need to set eyes')

    def set_eyes_from_angles(self, distances, elevations, azimuths):
        self._need_eye()
        self.transformer._eye = srf.get_points_from_angles(distances, elevations, azimuths)

    def set_eyes(self, eyes):
        self._need_eye()
        self.transformer._eye = eyes

    @property
    def eyes(self):
        return self.transformer._eye
