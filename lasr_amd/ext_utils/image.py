"""Silhouette transforms and the few OpenCV image operations the video loader needs, in numpy / scipy.

compute_dt, compute_dt_barrier, sample_contour: third_party/ext_utils/image.py:117-190 (under /root/reference/).
crop_pad / resize_*: what dataloader/vidbase.py:91-124 does through cv2.remap (integer maps = a crop with constant
padding) and cv2.resize (INTER_LINEAR: half-pixel centres, replicated border, no anti-aliasing; INTER_NEAREST:
floor(x * scale)).  OpenCV and scikit-image are not available here: restated, parity unpinned."""
import numpy as np


def compute_dt(mask, iters=10):
    """Distance to the (optionally dilated) mask, in units of the image size (image.py:117-125)."""
    from scipy.ndimage import binary_dilation, distance_transform_edt
    if iters > 1:
        mask = binary_dilation(mask.copy(), iterations=iters)
    return distance_transform_edt(1 - mask) / max(mask.shape)


def compute_dt_barrier(mask, k=50):
    from scipy.ndimage import distance_transform_edt
    diff = (distance_transform_edt(1 - mask) - distance_transform_edt(mask)) / max(mask.shape)
    return 1. / (1 + np.exp(k * -diff))


def contour_vertices(mask):
    """The vertices of skimage.measure.find_contours(mask, 0) for a 0 / 1 mask, as (row, col) pixel coordinates (image.py:146),
    one per crossed cell edge.  Marching squares places a vertex on every cell edge that joins a pixel above the level to one that
    is not, at from + (level - v_from) / (v_to - v_from) * (to - from); with level 0 and values in {0, 1} that fraction is 0 when
    the edge starts at the background pixel and 1 when it ends there: every vertex sits exactly ON the background pixel.  So the
    vertex LIST is: every background pixel once per foreground 4-neighbour inside the image (a background pixel in a concave
    corner of the silhouette appears two or three times, which weights sample_contour's draw the way the reference's concatenated
    contours do).  Not reproduced: find_contours closes a contour by repeating its first point -- one extra copy of one vertex
    per connected contour, at a position that depends on skimage's traversal order.  (skimage is absent from this image and
    restated from its published algorithm: parity unpinned; tests/test_dataloader.py enumerates the cell edges independently.)"""
    m = np.asarray(mask) > 0
    bg = ~m
    parts = []
    for fg_shifted, sl in ((m[:-1], np.s_[1:, :]), (m[1:], np.s_[:-1, :]), (m[:, :-1], np.s_[:, 1:]), (m[:, 1:], np.s_[:, :-1])):
        hit = np.zeros_like(m)
        hit[sl] = fg_shifted                                         # the neighbour above / below / left / right is foreground
        parts.append(np.argwhere(hit & bg))
    return np.concatenate(parts).astype(np.float64)


def sample_contour(mask, sample_size=1000, seed=None):
    """1000 points (x, y) in [-1, 1] sampled from a 2-pixel band around the silhouette boundary (image.py:140-190): the contour
    vertices (contour_vertices: find_contours at level 0), each shifted by the reference's 17 offsets and clipped to the image, then
    sample_size of them drawn without replacement (with replacement only when the band is smaller than that, where the reference's
    np.random.choice raises), x and y swapped, normalised.  The field is carried in the batch but not read by the loss path."""
    contour = contour_vertices(mask)                                 # (row, col)
    if len(contour) == 0:
        return np.zeros((sample_size, 2))
    size = np.asarray(mask).shape[0]
    offs = np.array([[0, 0], [0, 1], [0, 2], [0, -1], [0, -2], [1, 0], [2, 0], [-1, 0], [-2, 0], [-1, -1], [-2, -2],
                     [1, 1], [2, 2], [-1, 1], [-2, 2], [1, -1], [2, -2]])
    band = np.concatenate([np.clip(contour + o, 0, size - 1) for o in offs])
    rng = np.random.default_rng(seed)
    pick = band[rng.choice(len(band), sample_size, replace=len(band) < sample_size)]
    pick = (pick / size) * 2 - 1
    return pick[:, ::-1].copy()                                       # (x, y)


def crop_pad(src, x_start, y_start, size, border=0.):
    """src[y_start:y_start+size, x_start:x_start+size] with `border` outside the image: cv2.remap with the integer
    maps of vidbase.py:91-99 (linear and nearest interpolation coincide at integer coordinates)."""
    src = np.asarray(src)
    out_shape = (size, size) + src.shape[2:]
    out = np.empty(out_shape, src.dtype)
    out[...] = np.asarray(border, src.dtype) if np.ndim(border) else src.dtype.type(border)
    h, w = src.shape[:2]
    ys, xs = max(y_start, 0), max(x_start, 0)
    ye, xe = min(y_start + size, h), min(x_start + size, w)
    if ye > ys and xe > xs:
        out[ys - y_start:ye - y_start, xs - x_start:xe - x_start] = src[ys:ye, xs:xe]
    return out


def resize_linear(src, width, height):
    """cv2.resize(..., INTER_LINEAR): sample at (i + 0.5) * scale - 0.5 with the border replicated."""
    src = np.asarray(src, np.float64)
    h, w = src.shape[:2]

    def taps(n_out, n_in):
        pos = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(pos).astype(int)
        f = pos - i0
        return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), f
    y0, y1, fy = taps(height, h)
    x0, x1, fx = taps(width, w)
    fy = fy.reshape((-1, 1) + (1,) * (src.ndim - 2))
    fx = fx.reshape((1, -1) + (1,) * (src.ndim - 2))
    top = src[y0][:, x0] * (1 - fx) + src[y0][:, x1] * fx
    bot = src[y1][:, x0] * (1 - fx) + src[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def resize_nearest(src, width, height):
    """cv2.resize(..., INTER_NEAREST): source index floor(i * scale)."""
    src = np.asarray(src)
    h, w = src.shape[:2]
    ys = np.minimum(np.floor(np.arange(height) * (h / height)).astype(int), h - 1)
    xs = np.minimum(np.floor(np.arange(width) * (w / width)).astype(int), w - 1)
    return src[ys][:, xs]
