"""Wavefront .obj read/write (reference: soft_renderer/functional/load_obj.py:9-167, save_obj.py:44-87): geometry, per-vertex
colours, and -- for textured models such as the one scripts/render_syn.py:71 of the reference renders -- per-face surface
textures sampled from the material's atlas image by lasr_load_textures (the reference's load_textures CUDA extension).
Writing a texture atlas (save_obj with surface textures, create_texture_image) is off the path (OUT)."""
import os

import numpy as np
import torch


def parse_obj_materials(filename_obj):
    """The texture side of an .obj (load_obj.py:28-71, 9-25): -> (faces_uv [F,3,2] float32 in [0,1), material name per triangle,
    {material: Kd colour}, {material: texture file path}).  Triangles are the same fan triangulation as the geometry's; a corner
    without a texture index takes index 0, i.e. -1 after the 1-based shift -- the LAST vt entry, as in the reference; uv
    coordinates above 1 wrap (x % 1).  Raises if the file names no material library."""
    vts, tri, mats, mat, mtl = [], [], [], '', None
    with open(filename_obj) as f:
        lines = f.readlines()
    for line in lines:
        tok = line.split()
        if not tok:
            continue
        if tok[0] == 'vt':
            vts.append([float(v) for v in tok[1:3]])
        elif tok[0] == 'usemtl':
            mat = tok[1]
        elif tok[0] == 'mtllib':
            mtl = os.path.join(os.path.dirname(filename_obj), tok[1])
        elif tok[0] == 'f':
            idx = [int(t.split('/')[1]) if ('/' in t and '//' not in t) else 0 for t in tok[1:]]
            for i in range(len(idx) - 2):
                tri.append((idx[0], idx[i + 1], idx[i + 2]))
                mats.append(mat)
    if mtl is None:
        raise Exception('Failed to load textures.')                 # the reference's message (load_obj.py:143)
    vts = np.asarray(vts, np.float32).reshape(-1, 2)
    uv = vts[np.asarray(tri, np.int64).reshape(-1, 3) - 1]          # [F,3,2]; index -1 = last entry
    uv = np.where(uv > 1, uv % 1, uv).astype(np.float32)
    colors, files, name = {}, {}, ''
    with open(mtl) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == 'newmtl':
                name = tok[1]
            elif tok[0] == 'map_Kd':
                files[name] = os.path.join(os.path.dirname(filename_obj), tok[1])
            elif tok[0] == 'Kd':
                colors[name] = np.asarray([float(v) for v in tok[1:4]], np.float32)
    return uv, mats, colors, files


def _surface_textures(filename_obj, texture_res, device):
    """[F, R*R, 3] surface texels: ones, then each material's Kd colour, then each material's atlas sampled by the HIP kernel."""
    from PIL import Image
    from .load_textures import load_textures
    uv, mats, colors, files = parse_obj_materials(filename_obj)
    F = uv.shape[0]
    mats = np.asarray(mats)
    textures = torch.ones(F, texture_res ** 2, 3, dtype=torch.float32, device=device)
    for name, color in colors.items():
        sel = torch.from_numpy(mats == name).to(device)
        textures[sel] = torch.from_numpy(color).to(device)[None, None, :]
    faces_uv = torch.from_numpy(uv).to(device)
    for name, path in files.items():
        # grey, palette and RGBA atlases all become RGB (the reference stacks grey images and drops the alpha channel, :84-89)
        image = np.asarray(Image.open(path).convert('RGB')).astype(np.float32) / 255.
        image = np.ascontiguousarray(image[::-1])                   # v = 0 is the BOTTOM row of the picture
        upd = torch.from_numpy((mats == name).astype(np.int32)).to(device)
        sampled = load_textures(torch.from_numpy(image).to(device), faces_uv, texture_res, upd)
        sel = upd.bool()
        textures[sel] = sampled[sel]
    return textures


def load_obj(filename_obj, normalization=False, load_texture=False, texture_res=4, texture_type='surface',
             device=None):
    assert texture_type in ['surface', 'vertex']
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
    if load_texture and texture_type == 'surface':
        return vertices, faces, _surface_textures(filename_obj, texture_res, vertices.device)
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
