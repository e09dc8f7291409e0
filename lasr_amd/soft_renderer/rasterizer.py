"""Rasterisation stage (reference API: soft_renderer/rasterizer.py:9-55 under /root/reference/third_party/softras/)."""
import torch.nn as nn
import torch.nn.functional as F

from . import functional as srf

_CHOICES = (('dist_func', ('hard', 'euclidean', 'barycentric'), 'Distance function only support hard, euclidean and barycentric'),
            ('aggr_func_rgb', ('hard', 'softmax'), 'Aggregate function(rgb) only support hard and softmax'),
            ('aggr_func_alpha', ('hard', 'prod', 'sum'), 'Aggregate function(a) only support hard, prod and sum'),
            ('texture_type', ('surface', 'vertex'), 'Texture type only support surface and vertex'))


class SoftRasterizer(nn.Module):
    def __init__(self, image_size=256, background_color=[0, 0, 0], near=1, far=100,
                 anti_aliasing=False, fill_back=False, eps=1e-3,
                 sigma_val=1e-5, dist_func='euclidean', dist_eps=1e-4,
                 gamma_val=1e-4, aggr_func_rgb='softmax', aggr_func_alpha='prod',
                 texture_type='surface'):
        super().__init__()
        given = dict(dist_func=dist_func, aggr_func_rgb=aggr_func_rgb, aggr_func_alpha=aggr_func_alpha,
                     texture_type=texture_type)
        for name, allowed, msg in _CHOICES:
            if given[name] not in allowed:
                raise ValueError(msg)
        # plain attributes on purpose: LASR overwrites near/far/sigma_val/background_color between calls
        # (nnutils/mesh_net.py:306-316,356), near/far possibly with 0-dim device tensors
        self.image_size, self.background_color = image_size, background_color
        self.near, self.far = near, far
        self.anti_aliasing, self.fill_back, self.eps = anti_aliasing, fill_back, eps
        self.sigma_val, self.dist_func, self.dist_eps = sigma_val, dist_func, dist_eps
        self.gamma_val, self.aggr_func_rgb, self.aggr_func_alpha = gamma_val, aggr_func_rgb, aggr_func_alpha
        self.texture_type = texture_type

    def forward(self, mesh, mode=None):
        size = self.image_size * (2 if self.anti_aliasing else 1)
        images = srf.soft_rasterize(mesh.face_vertices, mesh.face_textures, size,
                                    self.background_color, self.near, self.far,
                                    self.fill_back, self.eps,
                                    self.sigma_val, self.dist_func, self.dist_eps,
                                    self.gamma_val, self.aggr_func_rgb, self.aggr_func_alpha,
                                    self.texture_type)
        if self.anti_aliasing:
            images = F.avg_pool2d(images, kernel_size=2, stride=2)
        return images
