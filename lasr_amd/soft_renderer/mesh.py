"""Batched triangle-mesh container (reference API: soft_renderer/mesh.py:9-175 under
/root/reference/third_party/softras/).  Derived quantities are cached and invalidated when
vertices or faces are reassigned, as the reference does with its *_update flags."""
import numpy as np
import torch

from . import functional as srf


def _to_device_tensor(x, dtype):
    if isinstance(x, np.ndarray):
        t = torch.from_numpy(x).to(dtype)
        return t.cuda() if torch.cuda.is_available() else t
    return x


class Mesh(object):
    def __init__(self, vertices, faces, textures=None, texture_res=1, texture_type='surface'):
        vertices = _to_device_tensor(vertices, torch.float32)
        faces = _to_device_tensor(faces, torch.int32)
        if vertices.ndimension() == 2:
            vertices = vertices[None]
        if faces.ndimension() == 2:
            faces = faces[None]
        self._vertices, self._faces = vertices, faces
        self.device = vertices.device
        self.texture_type = texture_type
        self.batch_size, self.num_vertices = vertices.shape[:2]
        self.num_faces = faces.shape[1]
        self._cache = {}
        self._fill_back = False

        if textures is None:
            if texture_type == 'surface':
                textures = torch.ones(self.batch_size, self.num_faces, texture_res ** 2, 3,
                                      dtype=torch.float32, device=self.device)
                self.texture_res = texture_res
            elif texture_type == 'vertex':
                textures = torch.ones(self.batch_size, self.num_vertices, 3, dtype=torch.float32, device=self.device)
                self.texture_res = 1
        else:
            textures = _to_device_tensor(textures, torch.float32)
            if textures.ndimension() == 3 and texture_type == 'surface':
                textures = textures[None]
            if textures.ndimension() == 2 and texture_type == 'vertex':
                textures = textures[None]
            self.texture_res = int(np.sqrt(textures.shape[2]))
        self._textures = textures
        self._origin = (self._vertices, self._faces, self._textures)

    # ---- attributes whose assignment invalidates the derived tensors
    @property
    def faces(self):
        return self._faces

    @faces.setter
    def faces(self, faces):
        self._faces = faces
        self.num_faces = faces.shape[1]
        self._cache.clear()

    @property
    def vertices(self):
        return self._vertices

    @vertices.setter
    def vertices(self, vertices):
        self._vertices = vertices
        self.num_vertices = vertices.shape[1]
        self._cache.clear()

    @property
    def textures(self):
        return self._textures

    @textures.setter
    def textures(self, textures):
        self._textures = textures

    def _cached(self, key, fn):
        if key not in self._cache:
            self._cache[key] = fn()
        return self._cache[key]

    @property
    def face_vertices(self):
        return self._cached('fv', lambda: srf.face_vertices(self.vertices, self.faces))

    @property
    def surface_normals(self):
        return self._cached('sn', lambda: srf.surface_normals(self.face_vertices))

    @property
    def vertex_normals(self):
        return self._cached('vn', lambda: srf.vertex_normals(self.vertices, self.faces))

    @property
    def face_textures(self):
        if self.texture_type == 'surface':
            return self.textures
        if self.texture_type == 'vertex':
            return srf.face_vertices(self.textures, self.faces)
        raise ValueError('texture type not applicable')

    def fill_back_(self):
        if not self._fill_back:
            self.faces = torch.cat((self.faces, self.faces[:, :, [2, 1, 0]]), dim=1)
            self.textures = torch.cat((self.textures, self.textures), dim=1)
            self._fill_back = True

    def reset_(self):
        self.vertices, self.faces, self.textures = self._origin
        self._fill_back = False

    @classmethod
    def from_obj(cls, filename_obj, normalization=False, load_texture=False, texture_res=1, texture_type='surface'):
        if load_texture:
            vertices, faces, textures = srf.load_obj(filename_obj, normalization=normalization,
                                                     texture_res=texture_res, load_texture=True,
                                                     texture_type=texture_type)
        else:
            vertices, faces = srf.load_obj(filename_obj, normalization=normalization,
                                           texture_res=texture_res, load_texture=False)
            textures = None
        return cls(vertices, faces, textures, texture_res, texture_type)

    def save_obj(self, filename_obj, save_texture=False, texture_res_out=16):
        if self.batch_size != 1:
            raise ValueError('Could not save when batch size >= 1')
        srf.save_obj(filename_obj, self.vertices[0], self.faces[0],
                     textures=self.textures[0] if save_texture else None,
                     texture_res=texture_res_out, texture_type=self.texture_type)
