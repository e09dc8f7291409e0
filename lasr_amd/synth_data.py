"""Synthetic spot3-like sequence rendered with this repo's own hard rasteriser (there is no
dataset on the GPU box).  Produces batches in exactly the layout LASRTrainer.set_input hands to the model
(/root/reference/nnutils/train_utils.py:125-181): same keys (trailing blanks included), pairs interleaved."""
import math

import numpy as np
import torch

from . import soft_renderer as sr
from . import synth

KEYS = ['input_imgs  ', 'imgs        ', 'masks       ', 'cams        ', 'depth_gt    ', 'flow        ',
        'dts_barrier ', 'ddts_barrier', 'mask_contour', 'pp          ', 'occ         ', 'oriimg_shape',
        'is_canonical', 'frameid', 'dataid']


def _hard_renderer(image_size):
    return sr.SoftRenderer(image_size=image_size, sigma_val=1e-6, camera_mode='look_at', perspective=False,
                           dist_func='hard', aggr_func_rgb='hard', aggr_func_alpha='hard', light_mode='vertex',
                           light_intensity_ambient=1., light_intensity_directionals=0., near=1., far=100.)


class SyntheticSequence:
    """n_frames views of the blobby mesh turning about y; frame pairs (i, i+dframe) with ground-truth flow."""

    def __init__(self, device, image_size=256, n_frames=3, nu=8, dframe=1, depth=10.0, focal=9.0):
        self.device, self.IS, self.n_frames, self.dframe = device, image_size, n_frames, dframe
        v, f, tex = synth.blobby_mesh(nu)
        V = v.shape[0]
        cam = np.empty((n_frames, V, 3), np.float32)
        for i in range(n_frames):
            c = (v @ synth.yaw_matrix(1.5 * math.pi + 2.0 * math.pi * i / n_frames).T).astype(np.float32)
            c[:, 2] += depth
            cam[i] = c
        cam_t = torch.from_numpy(cam).to(device)
        proj = torch.stack([cam_t[..., 0] * focal / cam_t[..., 2], cam_t[..., 1] * focal / cam_t[..., 2], cam_t[..., 2]], -1)
        faces = torch.from_numpy(f).to(device)[None].repeat(n_frames, 1, 1)
        r = _hard_renderer(image_size)
        eye = torch.tensor(r.transform.transformer._eye, device=device)[None, None]
        pre = (proj + eye) * proj.new_tensor([1, -1, 1])
        col = torch.from_numpy(tex).to(device)[None].repeat(n_frames, 1, 1)
        with torch.no_grad():
            rgba = r.render_mesh(sr.Mesh(pre, faces, textures=col, texture_type='vertex'))
            self.imgs = rgba[:, :3].contiguous()
            self.masks = (rgba[:, 3] > 0.5).float()
            # ground-truth flow i -> j: render the NDC position of the same surface point in frame j
            self.flow = {}
            for i in range(n_frames):
                for j in ((i + dframe) % n_frames, (i - dframe) % n_frames):
                    pos = r.render_mesh(sr.Mesh(pre[i:i + 1], faces[:1], textures=proj[j:j + 1], texture_type='vertex'))
                    here = r.render_mesh(sr.Mesh(pre[i:i + 1], faces[:1], textures=proj[i:i + 1], texture_type='vertex'))
                    fl = (pos[0, :2] - here[0, :2]) * self.masks[i][None]
                    self.flow[(i, j)] = torch.cat([fl, self.masks[i][None]], 0)
        from .ext_utils import image as image_utils
        m = (self.masks.cpu().numpy() > 0).astype(np.float64)
        # the two distance transforms of the loader (dataloader/vidbase.py:184-185): outside the silhouette, and
        # outside its 10-pixel dilation
        self.dts = torch.from_numpy(np.stack([image_utils.compute_dt(mm, iters=0) for mm in m]).astype(np.float32)).to(device)
        self.ddts = torch.from_numpy(np.stack([image_utils.compute_dt(mm, iters=10) for mm in m]).astype(np.float32)).to(device)
        self.mean = torch.tensor([0.485, 0.456, 0.406], device=device).view(1, 3, 1, 1)
        self.std = torch.tensor([0.229, 0.224, 0.225], device=device).view(1, 3, 1, 1)

    def pairs(self):
        return [(i, (i + self.dframe) % self.n_frames) for i in range(self.n_frames)]

    def packed(self):
        """The sequence as a PackedTable (dataloader/packed.py): row p = the batch dictionary of pair p, so that a batch is
        one gather launch into a persistent buffer."""
        from .dataloader.packed import PackedTable
        if getattr(self, '_packed', None) is None:
            self._packed = PackedTable([self._rows([p]) for p in range(len(self.pairs()))], self.device)
        return self._packed

    def batch(self, pair_ids):
        """pair_ids: B indices into pairs() (list, or an int64 device tensor).  Returns the dict LASR.forward expects."""
        if self.device.type == 'cuda' or torch.is_tensor(pair_ids):
            n = len(self.pairs())
            ids = pair_ids if torch.is_tensor(pair_ids) else torch.tensor([int(p) % n for p in pair_ids], dtype=torch.int64,
                                                                          device=self.device)
            return self.packed().gather(ids)                 # (the kernel clamps an out-of-range id to the last pair)
        return self._rows(pair_ids)

    def _rows(self, pair_ids):
        """The batch dictionary assembled with torch ops (builds the packed table; reference layout for the tests)."""
        prs = self.pairs()
        a = [prs[p % len(prs)][0] for p in pair_ids]
        b = [prs[p % len(prs)][1] for p in pair_ids]
        ids = a + b                                            # [frame t block ; frame t' block]
        B, IS, dev = len(pair_ids), self.IS, self.device
        imgs = self.imgs[ids]
        flow = torch.stack([self.flow[(i, j)] for i, j in list(zip(a, b)) + list(zip(b, a))])
        cams = torch.zeros(2 * B, 7, device=dev)
        cams[:, 0] = 1.0
        cams[:, 3] = 1.0
        out = {
            'input_imgs  ': (imgs - self.mean) / self.std,
            'imgs        ': imgs,
            'masks       ': self.masks[ids],
            'cams        ': cams,
            'depth_gt    ': torch.full((2 * B, 1), 10.0, device=dev),
            'flow        ': flow,
            'dts_barrier ': self.dts[ids][:, None],
            'ddts_barrier': self.ddts[ids][:, None],
            'mask_contour': torch.zeros(2 * B, 1, 1000, 2, device=dev),
            'pp          ': torch.zeros(2 * B, 2, device=dev),
            'occ         ': torch.ones(2 * B, IS, IS, device=dev),
            'oriimg_shape': torch.full((2 * B, 2), float(IS), device=dev),
            'is_canonical': torch.zeros(2 * B, device=dev),
            'frameid': torch.tensor(ids, device=dev, dtype=torch.float32),
            'dataid': torch.zeros(2 * B, device=dev),
        }
        # set_input's final interleave (train_utils.py:179-180); the model undoes it (mesh_net.py:155-156)
        return {k: v.view(2, B, -1).permute(1, 0, 2).reshape(v.shape) for k, v in out.items()}
