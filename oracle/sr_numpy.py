"""Independent numpy (fp64) restatement of the soft-rasteriser FORWARD pass -- TEST INFRASTRUCTURE ONLY.

Written separately from oracle/sr_oracle.c (vectorised over pixels, one face at a time, plain geometry instead of
the reference's barycentric-space edge projection where the two are mathematically the same) so that a slip in one
restatement shows up as a disagreement between the two.  Follows the semantics of
/root/reference/third_party/softras/soft_renderer/cuda/soft_rasterize_cuda_kernel.cu:308-483 ("K.cu"):
bbox reject, distance function, sigmoid, alpha aggregation BEFORE the depth test, clipped-barycentric depth,
z-buffer with lowest-index tie-break or online depth softmax, finalisation.  Vertex textures only.

Because it is fp64 and uses a different (better conditioned) distance formula it agrees with the fp64 instance of
the C oracle to ~1e-9 on well-conditioned meshes, not bit for bit.
"""
import numpy as np


def _pix(IS):
    c = (2.0 * np.arange(IS) + 1.0 - IS) / IS
    return np.meshgrid(c, c[::-1])                       # row 0 is the top (+y), K.cu:343-346


def _seg_closest(px, py, ax, ay, bx, by):
    ex, ey = bx - ax, by - ay
    t = ((px - ax) * ex + (py - ay) * ey) / max(ex * ex + ey * ey, 1e-300)
    t = np.clip(t, 0.0, 1.0)
    return ax + t * ex - px, ay + t * ey - py


def forward(face_vertices, textures, image_size, background_color=(0, 0, 0), near=1, far=100, fill_back=True,
            eps=1e-3, sigma_val=1e-5, dist_func='euclidean', dist_eps=1e-4, gamma_val=1e-4,
            aggr_func_rgb='softmax', aggr_func_alpha='prod', texture_type='vertex'):
    assert texture_type == 'vertex'
    fv = np.asarray(face_vertices, np.float64)
    tx = np.asarray(textures, np.float64)
    N, F = fv.shape[:2]
    IS = int(image_size)
    # scalar parameters are fp32 in the reference's kernels (K.cu:320-326)
    near, far, eps, sigma, gamma = (float(np.float32(v)) for v in (near, far, eps, sigma_val, gamma_val))
    thr = float(np.float32(np.float32(np.log(1.0 / dist_eps - 1.0)) * np.float32(sigma_val)))
    margin = float(np.sqrt(np.float32(thr)))
    X, Y = _pix(IS)
    out = np.zeros((N, 4, IS, IS))
    aggr = np.zeros((N, 2, IS, IS))
    np.seterr(over='ignore')                              # exp(+big) -> inf -> D = 0, as in C
    for n in range(N):
        alpha = np.ones((IS, IS)) if aggr_func_alpha == 'prod' else np.zeros((IS, IS))
        bg = np.asarray(background_color, np.float64)
        if aggr_func_rgb == 'softmax':
            ssum = np.full((IS, IS), float(np.exp(np.float32(eps) / np.float32(gamma))))
            smax = np.full((IS, IS), eps)
            rgb = bg[:, None, None] * ssum
        else:
            rgb = np.repeat(bg[:, None, None], IS, 1).repeat(IS, 2).astype(np.float64)
            zbest = np.full((IS, IS), 1e7)
            fbest = np.full((IS, IS), -1.0)
        for f in range(F):
            (x0, y0, z0), (x1, y1, z1), (x2, y2, z2) = fv[n, f]
            sel = ~((X > max(x0, x1, x2) + margin) | (X < min(x0, x1, x2) - margin) |
                    (Y > max(y0, y1, y2) + margin) | (Y < min(y0, y1, y2) - margin))
            if not sel.any():
                continue
            det = x2 * (y0 - y1) + x0 * (y1 - y2) + x1 * (y2 - y0)
            det = max(det, 1e-10) if det > 0 else min(det, -1e-10)
            w0 = ((y1 - y2) * X + (x2 - x1) * Y + (x1 * y2 - x2 * y1)) / det
            w1 = ((y2 - y0) * X + (x0 - x2) * Y + (x2 * y0 - x0 * y2)) / det
            w2 = ((y0 - y1) * X + (x1 - x0) * Y + (x0 * y1 - x1 * y0)) / det
            inside_open = (w0 > 0) & (w1 > 0) & (w2 > 0) & (w0 < 1) & (w1 < 1) & (w2 < 1)
            inside_closed = (w0 >= 0) & (w1 >= 0) & (w2 >= 0) & (w0 <= 1) & (w1 <= 1) & (w2 <= 1)
            if dist_func == 'hard':
                D = inside_closed.astype(np.float64)
                keep = sel & inside_closed
            elif dist_func == 'barycentric':
                d = np.minimum(np.minimum(w0, w1), w2)
                dis = np.where(d > 0, d * d, -(d * d))
                keep = sel & ~(-dis >= thr)
                D = 1.0 / (1.0 + np.exp(-dis / sigma))
            else:
                cand = [_seg_closest(X, Y, x0, y0, x1, y1), _seg_closest(X, Y, x1, y1, x2, y2),
                        _seg_closest(X, Y, x2, y2, x0, y0)]
                d2 = np.stack([cx * cx + cy * cy for cx, cy in cand])
                dis = d2.min(0)
                # inside: +distance to the nearest edge LINE; outside: -distance to the triangle (K.cu:69-150)
                if inside_open.any():
                    lines = []
                    for (ax, ay, bx, by) in ((x0, y0, x1, y1), (x1, y1, x2, y2), (x2, y2, x0, y0)):
                        ex, ey = bx - ax, by - ay
                        lines.append(((X - ax) * ey - (Y - ay) * ex) ** 2 / max(ex * ex + ey * ey, 1e-300))
                    dis = np.where(inside_open, np.stack(lines).min(0), dis)
                sign = np.where(inside_open, 1.0, -1.0)
                keep = sel & ~((sign < 0) & (dis >= thr))
                D = 1.0 / (1.0 + np.exp(-sign * dis / sigma))
            if not keep.any():
                continue
            if aggr_func_alpha == 'hard':
                alpha = np.where(keep & (D > 0.5), 1.0, alpha)
            elif aggr_func_alpha == 'sum':
                alpha = alpha + np.where(keep, D, 0.0)
            else:
                alpha = alpha * np.where(keep, 1.0 - D, 1.0)
            c0, c1, c2 = (np.clip(w, 0.0, 1.0) for w in (w0, w1, w2))
            s = np.maximum(c0 + c1 + c2, 1e-5)
            c0, c1, c2 = c0 / s, c1 / s, c2 / s
            zp = 1.0 / (c0 / z0 + c1 / z1 + c2 / z2)
            keep = keep & ~((zp < near) | (zp > far))
            front = (y2 - y0) * (x1 - x0) < (y1 - y0) * (x2 - x0)
            col = [c0 * tx[n, f, 0, k] + c1 * tx[n, f, 1, k] + c2 * tx[n, f, 2, k] for k in range(3)]
            if aggr_func_rgb == 'hard':
                win = keep & (zp < zbest) & inside_closed & (fill_back or front)
                zbest = np.where(win, zp, zbest)
                fbest = np.where(win, float(f), fbest)
                for k in range(3):
                    rgb[k] = np.where(win, col[k], rgb[k])
            elif fill_back or front:
                zn = (far - zp) / (far - near)
                newmax = keep & (zn > smax)
                resc = np.where(newmax, np.exp((smax - zn) / gamma), 1.0)
                smax = np.where(newmax, zn, smax)
                ez = np.exp((zn - smax) / gamma)
                ssum = np.where(keep, resc * ssum + ez * D, ssum)
                for k in range(3):
                    rgb[k] = np.where(keep, resc * rgb[k] + ez * D * col[k], rgb[k])
        if aggr_func_alpha == 'hard':
            out[n, 3] = alpha
        elif aggr_func_alpha == 'sum':
            out[n, 3] = alpha / F
        else:
            out[n, 3] = 1.0 - alpha
        if aggr_func_rgb == 'hard':
            out[n, :3] = rgb
            aggr[n, 0], aggr[n, 1] = zbest, fbest
        else:
            out[n, :3] = rgb / ssum
            aggr[n, 0], aggr[n, 1] = ssum, smax
    return dict(soft_colors=out, aggrs_info=aggr)
