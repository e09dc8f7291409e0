"""Deterministic synthetic inputs for parity tests and bench.py.

There is no dataset on the GPU box, so every workload is generated here:
  M1  geodesic icosahedron nu=8   (V=642,  F=1280)  -- spot3 stage-0 mesh size
  M2  geodesic icosahedron nu=11  (V=1212, F=2420)  -- the "~1.2k vert / 2.3k face" mesh
shape  = unit sphere * diag(0.7, 0.45, 0.5) + 0.05 * N(0,1)    (default_rng(0))
colour = U(0,1) per vertex                                       (default_rng(1))
frame i of Nf: yaw 3pi/2 + 2pi i/Nf about y (cf. /root/reference/scripts/render_syn.py:145),
depth 10, focal 9, principal point 0  ->  coverage of roughly a quarter of a 256x256 image.
"""
import math

import numpy as np


def icosahedron():
    p = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, p, 0], [1, p, 0], [-1, -p, 0], [1, -p, 0],
                  [0, -1, p], [0, 1, p], [0, -1, -p], [0, 1, -p],
                  [p, 0, -1], [p, 0, 1], [-p, 0, -1], [-p, 0, 1]], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
                  [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
                  [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], np.int64)
    return v, f


def geodesic_sphere(nu):
    """Class-I geodesic icosahedron of frequency nu: V = 10 nu^2 + 2, F = 20 nu^2."""
    bv, bf = icosahedron()
    index = {}
    verts = []

    def vid(p):
        key = tuple(np.round(p, 7))
        if key not in index:
            index[key] = len(verts)
            verts.append(p)
        return index[key]

    faces = []
    for a, b, c in bf:
        A, B, C = bv[a], bv[b], bv[c]
        grid = {}
        for i in range(nu + 1):
            for j in range(nu + 1 - i):
                grid[(i, j)] = vid((A * (nu - i - j) + B * i + C * j) / nu)
        for i in range(nu):
            for j in range(nu - i):
                faces.append((grid[(i, j)], grid[(i + 1, j)], grid[(i, j + 1)]))
                if i + j < nu - 1:
                    faces.append((grid[(i + 1, j)], grid[(i + 1, j + 1)], grid[(i, j + 1)]))
    v = np.asarray(verts)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v.astype(np.float32), np.asarray(faces, np.int64)


def blobby_mesh(nu=8):
    v, f = geodesic_sphere(nu)
    rng = np.random.default_rng(0)
    v = v.astype(np.float64) * np.array([0.7, 0.45, 0.5]) + 0.05 * rng.standard_normal(v.shape)
    tex = np.random.default_rng(1).uniform(0, 1, v.shape)
    return v.astype(np.float32), f, tex.astype(np.float32)


def yaw_matrix(theta):
    c, s = math.cos(theta), math.sin(theta)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)


def frame_vertices(v, n_frames, depth=10.0, focal=9.0, first=0, count=None):
    """Camera-space + pinhole-projected + y-flipped vertices, [count, V, 3] float32 (x f/z, -y f/z, z)."""
    count = n_frames if count is None else count
    out = np.empty((count, v.shape[0], 3), np.float32)
    for k in range(count):
        i = (first + k) % n_frames
        R = yaw_matrix(1.5 * math.pi + 2.0 * math.pi * i / n_frames)
        cam = (v @ R.T).astype(np.float32)
        cam[:, 2] += np.float32(depth)
        out[k, :, 0] = cam[:, 0] * np.float32(focal) / cam[:, 2]
        out[k, :, 1] = -(cam[:, 1] * np.float32(focal) / cam[:, 2])
        out[k, :, 2] = cam[:, 2]
    return out


def near_far(z):
    """near/far rule of /root/reference/nnutils/mesh_net.py:304-311."""
    zmin, zmax = float(np.min(z)), float(np.max(z))
    return zmin - (zmax - zmin) / 2, zmax + (zmax - zmin) / 2


def raster_batch(nu=8, n_frames=3, count=None, first=0):
    """face_vertices [n,F,3,3], face_textures [n,F,3,3] (vertex colours), near, far."""
    v, f, tex = blobby_mesh(nu)
    pv = frame_vertices(v, n_frames, first=first, count=count)
    near, far = near_far(frame_vertices(v, n_frames)[:, :, 2])
    fv = pv[:, f]                      # [n, F, 3, 3]
    ft = np.broadcast_to(tex[f][None], fv.shape).copy()
    return fv, ft, near, far


def upstream_grad(n, image_size, seed=2):
    rng = np.random.default_rng(seed)
    P = image_size * image_size
    return (rng.standard_normal((n, 4, image_size, image_size)) / P).astype(np.float32)


# the raster configuration LASR trains with (/root/reference/nnutils/mesh_net.py:136-145)
LASR_MODES = dict(background_color=(1, 1, 1), fill_back=True, eps=1e-3, sigma_val=1e-4, dist_func='euclidean',
                  dist_eps=1e-4, gamma_val=1e-2, aggr_func_rgb='softmax', aggr_func_alpha='prod',
                  texture_type='vertex')


def label_palette(n):
    """n distinct RGB colours in 0..255 for the part visualisation (the reference uses a Cityscapes colour map,
    nnutils/geom_utils.py:97-254; only used for logging)."""
    out = []
    for i in range(n):
        h = (i * 0.61803398875) % 1.0
        k = int(h * 6)
        f = h * 6 - k
        q, t = 1 - f, f
        r, g, b = [(1, t, 0), (q, 1, 0), (0, 1, t), (0, q, 1), (t, 0, 1), (1, 0, q)][k % 6]
        out.append([255 * r, 255 * g, 255 * b])
    return np.asarray(out, np.float32)
