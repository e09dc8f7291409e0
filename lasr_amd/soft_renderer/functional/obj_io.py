"""Wavefront .obj read/write for checkpoints (reference: soft_renderer/functional/load_obj.py:104-167,
save_obj.py:44-87).  Geometry and per-vertex colours only: the texture-atlas paths of the reference go
through two CUDA extensions that are off the hot path (SURVEY.md section 2, rows marked OUT)."""
import os

import numpy as np
import torch


def load_obj(filename_obj, normalization=False, load_texture=False, texture_res=4, texture_type='surface',
             device=None):
    assert texture_type in ['surface', 'vertex']
    if load_texture and texture_type == 'surface':
        raise NotImplementedError('surface texture atlases are outside the hot path (SURVEY.md section 2)')
    device = device or ('cuda' if torch.cuda.is_available() else 'cpu')
    verts, cols, faces = [], [], []
    with open(filename_obj) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == 'v':
                verts.append([float(v) for v in tok[1:4]])
                cols.append([float(v) for v in tok[4:7]])
            elif tok[0] == 'f':
                ids = [int(t.split('/')[0]) for t in tok[1:]]
                for i in range(len(ids) - 2):                     # fan triangulation
                    faces.append((ids[0], ids[i + 1], ids[i + 2]))
    vertices = torch.from_numpy(np.asarray(verts, np.float32)).to(device)
    faces = torch.from_numpy(np.asarray(faces, np.int32)).to(device) - 1
    if normalization:                                              # unit cube centred at zero
        vertices -= vertices.min(0)[0][None, :]
        vertices /= torch.abs(vertices).max()
        vertices *= 2
        vertices -= vertices.max(0)[0][None, :] / 2
    if load_texture:
        return vertices, faces, torch.from_numpy(np.asarray(cols, np.float32)).to(device)
    return vertices, faces


def save_obj(filename, vertices, faces, textures=None, texture_res=16, texture_type='surface'):
    assert vertices.ndimension() == 2 and faces.ndimension() == 2
    assert texture_type in ['surface', 'vertex']
    if textures is not None and texture_type == 'surface':
        raise NotImplementedError('surface texture atlases are outside the hot path (SURVEY.md section 2)')
    v = vertices.detach().cpu().numpy()
    fc = faces.detach().cpu().numpy()
    tx = textures.detach().cpu().numpy() if textures is not None else None
    with open(filename, 'w') as f:
        f.write('# %s\n#\n\n' % os.path.basename(filename))
        for i, p in enumerate(v):
            if tx is not None:
                f.write('v %.8f %.8f %.8f %.8f %.8f %.8f\n' % (p[0], p[1], p[2], tx[i, 0], tx[i, 1], tx[i, 2]))
            else:
                f.write('v %.8f %.8f %.8f\n' % (p[0], p[1], p[2]))
        f.write('\n')
        for t in fc:
            f.write('f %d %d %d\n' % (t[0] + 1, t[1] + 1, t[2] + 1))
