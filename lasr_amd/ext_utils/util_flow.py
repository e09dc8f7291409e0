"""PFM files, the container of the optical-flow / occlusion maps (reference: third_party/ext_utils/util_flow.py:36-131
under /root/reference/).  Rows are stored bottom-up; a negative scale marks little-endian data."""
import re

import numpy as np


def readPFM(path):
    """-> (array [H,W] or [H,W,3] float32 in top-down row order, scale)."""
    with open(path, 'rb') as fh:
        header = fh.readline().rstrip().decode('utf-8')
        if header not in ('PF', 'Pf'):
            raise Exception('Not a PFM file.')
        dims = re.match(r'^(\d+)\s(\d+)\s$', fh.readline().decode('utf-8'))
        if not dims:
            raise Exception('Malformed PFM header.')
        width, height = map(int, dims.groups())
        scale = float(fh.readline().rstrip().decode('utf-8'))
        endian = '<' if scale < 0 else '>'
        data = np.frombuffer(fh.read(), dtype=endian + 'f4')
    shape = (height, width, 3) if header == 'PF' else (height, width)
    return np.flipud(data.reshape(shape)).astype(np.float32), abs(scale)


def write_pfm(path, image, scale=1):
    if image.dtype.name != 'float32':
        raise Exception('Image dtype must be float32.')
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise Exception('Image must have H x W x 3, H x W x 1 or H x W dimensions.')
    image = np.flipud(image)
    little = image.dtype.byteorder == '<' or (image.dtype.byteorder == '=' and np.little_endian)
    with open(path, 'wb') as fh:
        fh.write(b'PF\n' if color else b'Pf\n')
        fh.write(('%d %d\n' % (image.shape[1], image.shape[0])).encode())
        fh.write(('%f\n' % (-scale if little else scale)).encode())
        fh.write(np.ascontiguousarray(image).tobytes())
