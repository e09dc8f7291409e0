"""Image / silhouette / flow pair loader (reference: /root/reference/dataloader/vidbase.py:38-231).

Same element dictionary as the reference's BaseDataset.__getitem__.  Differences, all behind the same interface:
* OpenCV is replaced by PIL (decode) and the numpy restatements of lasr_amd/ext_utils/image.py (crop, resize);
* every distinct (frame, neighbour) pair is decoded, cropped, resized and distance-transformed ONCE and kept
  (SURVEY.md section 8 row f2): the reference repeats its pair list ~200/len times per epoch (vid.py:78-80) and redoes
  the JPEG/PFM decode, two EDTs and a contour trace for every repetition.
"""
import os.path as osp

import numpy as np
from torch.utils.data import Dataset

from ..ext_utils import image as image_utils
from ..ext_utils.util_flow import readPFM


def _read_rgb(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'), np.float64) / 255.0


def _read_gray(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('L'))


class BaseDataset(Dataset):
    """img, mask, flow data loader"""

    def __init__(self, opts, filter_key=None):
        self.opts = opts
        self.img_size = opts.img_size
        self._cache = {}

    def __len__(self):
        return self.num_imgs

    # ---- one side of a pair ------------------------------------------------------------------
    def _load_mask(self, idx, img, erosions):
        mask = _read_gray(self.masklist[idx])
        if mask.shape[0] != img.shape[0] or mask.shape[1] != img.shape[1]:      # vidbase.py:68-70, 74-76
            from scipy.ndimage import binary_erosion
            mask = image_utils.resize_nearest(mask, img.shape[1], img.shape[0])
            mask = binary_erosion(mask, iterations=erosions)
        return np.asarray(mask)

    @staticmethod
    def _box(mask):
        """Square crop box around the silhouette, 1.2 x its larger half-extent (vidbase.py:98-107)."""
        ys, xs = np.where(mask > 0)
        center = ((xs.max() + xs.min()) // 2, (ys.max() + ys.min()) // 2)
        half = int(1.2 * max((xs.max() - xs.min()) // 2, (ys.max() - ys.min()) // 2))
        return center, half

    def _crop_resize(self, img, mask, flow, occ, color, center, half):
        size = self.opts.img_size
        xs, ys = int(center[0] - half), int(center[1] - half)
        img = image_utils.crop_pad(img, xs, ys, 2 * half, border=color)
        mask = image_utils.crop_pad(mask.astype(np.int64), xs, ys, 2 * half)
        flow = image_utils.crop_pad(flow, xs, ys, 2 * half)
        occ = image_utils.crop_pad(occ, xs, ys, 2 * half)
        return (image_utils.resize_linear(img, size, size), image_utils.resize_nearest(mask, size, size),
                image_utils.resize_linear(flow, size, size), image_utils.resize_linear(occ, size, size))

    def __getitem__(self, index):
        im0idx = self.baselist[index]
        forward = self.directlist[index] == 1
        im1idx = im0idx + self.dframe if forward else im0idx - self.dframe
        key = (im0idx, im1idx)
        elem = self._cache.get(key)
        if elem is None:
            elem = self._cache[key] = self._build(im0idx, im1idx, forward)
        elem = dict(elem)
        elem['inds'] = elem['indsn'] = index
        return elem

    def _build(self, im0idx, im1idx, forward):
        size = self.opts.img_size
        img, imgn = _read_rgb(self.imglist[im0idx]), _read_rgb(self.imglist[im1idx])
        shape = img.shape
        mask = self._load_mask(im0idx, img, 2)
        maskn = self._load_mask(im1idx, imgn, 1)
        fg, fgn = mask > 0, maskn > 0
        # background painted with the complement of the mean foreground colour (vidbase.py:78-82)
        color = 1 - img[fg].mean(0)
        colorn = 1 - imgn[fgn].mean(0)
        img = img * fg[..., None] + color * (1 - fg[..., None])
        imgn = imgn * fgn[..., None] + colorn * (1 - fgn[..., None])

        if forward:                                                            # vidbase.py:84-95
            flowpath, flowpathn = self.flowfwlist[im0idx], self.flowbwlist[im0idx + self.dframe]
        else:
            flowpath, flowpathn = self.flowbwlist[im0idx], self.flowfwlist[im0idx - self.dframe]
        flow, flown = readPFM(flowpath)[0], readPFM(flowpathn)[0]
        occ = readPFM(flowpath.replace('flo-', 'occ-'))[0]
        occn = readPFM(flowpathn.replace('flo-', 'occ-'))[0]

        center, half = self._box(mask)
        centern, halfn = self._box(maskn)
        img, mask, flow, occ = self._crop_resize(img, mask, flow, occ, color, center, half)
        imgn, maskn, flown, occn = self._crop_resize(imgn, maskn, flown, occn, colorn, centern, halfn)

        # flow between the two crops, in units of the resized target frame, then in [-1,1] (vidbase.py:127-147)
        alp, alpn = 2 * half / size, 2 * halfn / size
        betax, betay = np.meshgrid(range(size), range(size))
        x0, y0, x0n, y0n = center[0] - half, center[1] - half, centern[0] - halfn, centern[1] - halfn

        def rebase(fl, oc, dx, dy, da, a_to):
            fl = fl.copy()
            fl[:, :, 0] += dx + betax * da
            fl[:, :, 1] += dy + betay * da
            fl /= a_to
            fl[:, :, 0] = 2 * (fl[:, :, 0] / size)
            fl[:, :, 1] = 2 * (fl[:, :, 1] / size)
            fl[:, :, 2] = np.logical_and(fl[:, :, 2] != 0, oc < 10)             # valid pixels
            return fl
        flow = rebase(flow, occ, x0 - x0n, y0 - y0n, alp - alpn, alpn)
        flown = rebase(flown, occn, x0n - x0, y0n - y0, alpn - alp, alp)

        cam = np.asarray([1., 0., 0., 1., 0., 0., 0.])
        camn = np.asarray([1., 0., 0., 1., 0., 0., 0.])
        depth, depthn = 0., 0.
        if osp.exists(self.camlist[im0idx]):                                    # vidbase.py:169-180
            cam0 = np.loadtxt(self.camlist[im0idx]).astype(np.float32)
            cam1 = np.loadtxt(self.camlist[im1idx]).astype(np.float32)
            cam[:], camn[:] = cam0[:-1], cam1[:-1]
            depth, depthn = cam0[-1:], cam1[-1:]
        cam[0], camn[0] = 1. / alp, 1. / alpn                                   # focal length follows the rescale

        masks = np.stack([(mask > 0).astype(float), (maskn > 0).astype(float)])
        return {
            'img': np.transpose(img, (2, 0, 1)), 'imgn': np.transpose(imgn, (2, 0, 1)),
            'mask': masks,
            'mask_dts': np.stack([image_utils.compute_dt(m, iters=0) for m in masks]),
            'dmask_dts': np.stack([image_utils.compute_dt(m, iters=10) for m in masks]),
            'mask_contour': np.stack([image_utils.sample_contour(m, seed=im0idx * 131 + k) for k, m in enumerate(masks)]),
            'cam': cam, 'camn': camn,
            'flow': np.transpose(flow, (2, 0, 1)), 'flown': np.transpose(flown, (2, 0, 1)),
            'pps': np.stack([np.asarray([float(x0), float(y0)]), np.asarray([float(x0n), float(y0n)])]),
            'depth': depth, 'depthn': depthn,
            'is_canonical': self.can_frame == im0idx, 'is_canonicaln': self.can_frame == im1idx,
            'dataid': getattr(self, 'dataid', 0),
            'id0': im0idx, 'id1': im1idx,
            'occ': occ, 'occn': occn,
            'shape': np.asarray(shape)[:2][::-1].copy(),
        }
