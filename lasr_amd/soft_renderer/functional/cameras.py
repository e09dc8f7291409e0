"""Camera transforms of the SoftRas operator (reference: soft_renderer/functional/{look_at,look,
orthogonal,perspective,projection,get_points_from_angles}.py under /root/reference/third_party/softras/)."""
import math

import numpy as np
import torch
import torch.nn.functional as F


_CONSTS = {}


def const_tensor(values, device):
    """Device tensor for a small Python constant (eye position, light colour, axis flips ...), created once per
    (device, value): building it anew on every call is a pageable host->device copy, which costs a launch each time
    and is illegal while a HIP graph is being captured."""
    key = (str(device), tuple(float(v) for v in values))
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(list(key[1]), dtype=torch.float32, device=device)
    return t


def _vec(x, device, batch):
    """list/tuple/ndarray/tensor -> float32 tensor [batch or 1, 3] on `device`."""
    if isinstance(x, (list, tuple)):
        x = const_tensor(x, device)
    elif isinstance(x, np.ndarray):
        x = torch.from_numpy(x).to(device)
    else:
        x = x.to(device)
    if x.ndimension() == 1:
        x = x[None, :].repeat(batch, 1)
    return x


def _basis(z_axis, up):
    z_axis = F.normalize(z_axis, eps=1e-5)
    x_axis = F.normalize(torch.cross(up, z_axis, dim=-1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=-1), eps=1e-5)
    return torch.stack((x_axis, y_axis, z_axis), dim=1)       # rows = camera axes, [B,3,3]


def _apply(vertices, eye, r):
    if vertices.shape != eye.shape:
        eye = eye[:, None, :]
    return torch.matmul(vertices - eye, r.transpose(1, 2))


_LOOK_AT = {}


def _const_look_at(eye, at, up, device):
    """Constant camera: the basis is built once per (eye, at, up, device) with the same ops as the general path and
    reused: -> (eye [1,1,3], R^T [3,3] or None when R is exactly the identity, e.g. LASR's eye on the -z axis)."""
    key = (str(device), tuple(map(float, eye)), tuple(map(float, at)), tuple(map(float, up)))
    hit = _LOOK_AT.get(key)
    if hit is None:
        e, a, u = _vec(list(eye), device, 1), _vec(list(at), device, 1), _vec(list(up), device, 1)
        r = _basis(a - e, u)[0]
        identity = bool(torch.equal(r.cpu(), torch.eye(3)))            # (v - eye) @ I == v - eye bit for bit
        hit = _LOOK_AT[key] = (e[:, None, :].clone(), None if identity else r.t().contiguous())
    return hit


def look_at(vertices, eye, at=[0, 0, 0], up=[0, 1, 0]):
    """World -> camera looking from `eye` towards `at` (look_at.py:6-62)."""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    if all(isinstance(x, (list, tuple)) for x in (eye, at, up)):
        e, rt = _const_look_at(eye, at, up, vertices.device)
        return vertices - e if rt is None else torch.matmul(vertices - e, rt)
    bs, dev = vertices.shape[0], vertices.device
    eye, at, up = _vec(eye, dev, bs), _vec(at, dev, bs), _vec(up, dev, bs)
    return _apply(vertices, eye, _basis(at - eye, up))


def look(vertices, eye, direction=[0, 1, 0], up=[0, 1, 0]):
    """World -> camera at `eye` looking along `direction`.  (The reference version, look.py:6-52,
    dereferences its `up=None` default and cannot run; this one takes the usual +y up.)"""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    bs, dev = vertices.shape[0], vertices.device
    eye, direction, up = _vec(eye, dev, bs), _vec(direction, dev, bs), _vec(up, dev, bs)
    return _apply(vertices, eye, _basis(direction, up))


def orthogonal(vertices, scale):
    """x,y scaled, z kept (orthogonal.py:4-17)."""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    if isinstance(scale, (int, float)):
        if scale == 1:
            return vertices                                           # x * 1 is x: nothing to launch
        return vertices * const_tensor([scale, scale, 1], vertices.device)
    return torch.stack((vertices[:, :, 0] * scale, vertices[:, :, 1] * scale, vertices[:, :, 2]), dim=2)


def perspective(vertices, angle=30.):
    """x,y divided by z and by tan(angle) (perspective.py:5-21)."""
    if vertices.ndimension() != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    width = torch.tan(torch.tensor(angle / 180 * math.pi, dtype=torch.float32, device=vertices.device))[None, None]
    z = vertices[:, :, 2]
    return torch.stack((vertices[:, :, 0] / z / width, vertices[:, :, 1] / z / width, z), dim=2)


def projection(vertices, P, dist_coeffs, orig_size):
    """3x4 projection with OpenCV-style radial/tangential distortion (projection.py:4-36)."""
    hom = torch.cat([vertices, torch.ones_like(vertices[:, :, :1])], dim=-1)
    cam = torch.bmm(hom, P.transpose(2, 1))
    x, y, z = cam[:, :, 0], cam[:, :, 1], cam[:, :, 2]
    x_ = x / (z + 1e-5)
    y_ = y / (z + 1e-5)
    k1, k2, p1, p2, k3 = (dist_coeffs[:, None, i] for i in range(5))
    r = torch.sqrt(x_ ** 2 + y_ ** 2)
    radial = 1 + k1 * (r ** 2) + k2 * (r ** 4) + k3 * (r ** 6)
    xd = x_ * radial + 2 * p1 * x_ * y_ + p2 * (r ** 2 + 2 * x_ ** 2)
    yd = y_ * radial + p1 * (r ** 2 + 2 * y_ ** 2) + 2 * p2 * x_ * y_
    xd = 2 * (xd - orig_size / 2.) / orig_size
    yd = 2 * (yd - orig_size / 2.) / orig_size
    return torch.stack([xd, yd, z], dim=-1)


def get_points_from_angles(distance, elevation, azimuth, degrees=True):
    """Eye position on a sphere (get_points_from_angles.py:5-25)."""
    if isinstance(distance, (float, int)):
        if degrees:
            elevation, azimuth = math.radians(elevation), math.radians(azimuth)
        return (distance * math.cos(elevation) * math.sin(azimuth),
                distance * math.sin(elevation),
                -distance * math.cos(elevation) * math.cos(azimuth))
    if degrees:
        elevation = math.pi / 180. * elevation
        azimuth = math.pi / 180. * azimuth
    return torch.stack([distance * torch.cos(elevation) * torch.sin(azimuth),
                        distance * torch.sin(elevation),
                        -distance * torch.cos(elevation) * torch.cos(azimuth)]).transpose(1, 0)
