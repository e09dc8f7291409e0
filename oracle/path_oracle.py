"""CPU restatements (torch, eager) of the Python pieces of the LASR hot path -- TEST INFRASTRUCTURE ONLY.

Each function follows the cited reference lines operation by operation (paths relative to /root/reference/)
so torch autograd reproduces the reference gradients.  Pinned by tests/test_path_oracle.py against fixtures
captured from the imported reference (obj_to_cam, pinhole_cam, Laplacian, Flatten, ARAP); the loss tables
restate nnutils/mesh_net.py, which cannot be imported here (absl / kornia / pytorch3d are missing) and has no
tests upstream: "parity unpinned" for those three, covered by hand-computable cases instead.
"""
import numpy as np
import torch


def obj_to_cam(verts, Rmat, Tmat, nmesh, n_hypo, skin, tocam=True):
    """nnutils/geom_utils.py:45-71."""
    verts = verts.reshape(-1, verts.shape[1], 3)
    bodyR, bodyT = Rmat[::nmesh], Tmat[::nmesh]
    if nmesh > 1:
        vs = []
        for k in range(nmesh - 1):
            partR, partT = Rmat[k + 1::nmesh], Tmat[k + 1::nmesh]
            vs.append((verts.matmul(partR) + partT)[:, None])
        vs = (torch.cat(vs, 1) * skin).sum(1)
    else:
        vs = verts
    return vs.matmul(bodyR) + bodyT if tocam else vs


def pinhole_cam(verts, pp, fl):
    """nnutils/geom_utils.py:27-34."""
    n_hypo = verts.shape[0] // pp.shape[0]
    pp = pp[:, None].repeat(1, n_hypo, 1).view(-1, 2)
    fl = fl[:, None].reshape(-1, 1)
    y = pp[:, 1:2] + verts[:, :, 1] * fl / verts[:, :, 2]
    x = pp[:, 0:1] + verts[:, :, 0] * fl / verts[:, :, 2]
    return torch.stack([x, y, verts[:, :, 2], verts[:, :, 3]], -1)


def mask_loss_table(mask_pred, masks, occ):
    """nnutils/mesh_net.py:374-388; mask_pred [I,H,S,S], masks/occ [I,S,S] -> [I,H]."""
    sub = (mask_pred - masks[:, None]).pow(2)
    out = torch.zeros(mask_pred.shape[0], mask_pred.shape[1], dtype=mask_pred.dtype)
    for i in range(out.shape[0]):
        for j in range(out.shape[1]):
            out[i, j] = sub[i, j][occ[i] != 0].mean()
    return 0.5 * out


def flow_loss_table(flow_rd, flow_obs, bgmask, occ, masks):
    """nnutils/mesh_net.py:393-413; flow_rd [I,H,S,S,2], flow_obs [I,>=2,S,S], bgmask [I,H,S,S] bool -> ([I,H], map)."""
    I, H = flow_rd.shape[:2]
    mask = (~bgmask) & ((occ != 0)[:, None] & (masks > 0)[:, None]).repeat(1, H, 1, 1)
    fmap = torch.norm(flow_rd - flow_obs[:, None, :2].permute(0, 1, 3, 4, 2), 2, -1)
    w = (-occ).sigmoid()[:, None].repeat(1, H, 1, 1)
    w = torch.stack([w[i] / w[i][mask[i]].mean() for i in range(I)])
    fmap = fmap * w
    out = torch.zeros(I, H, dtype=flow_rd.dtype)
    for i in range(I):
        for j in range(H):
            out[i, j] = fmap[i, j][mask[i, j]].mean() if mask[i, j].sum() > 0 else 0.
    return 0.5 * out, fmap


def tex_loss_table(img_obs, img_white, texture_render, fgmask, occ, l1tex_wt=1.0):
    """nnutils/mesh_net.py:425-441 without the perceptual term; texture_render [I,H,3,S,S], fgmask [I,H,S,S]."""
    I, H = texture_render.shape[:2]
    img_rnd = texture_render * fgmask[:, :, None]
    out = torch.zeros(I, H, dtype=texture_render.dtype)
    for i in range(I):
        for j in range(H):
            a = (img_obs[i] - img_rnd[i, j]).abs().mean(0)[occ[i] != 0].mean()
            b = (img_white[i] - texture_render[i, j]).abs().mean(0)[occ[i] != 0].mean()
            out[i, j] = (a + b) * 2 * l1tex_wt
    return out


def _adjacency(nv, faces):
    """0/1 dense adjacency as nnutils/loss_utils.py:36-43 builds it."""
    A = np.zeros([nv, nv], np.float32)
    f = np.asarray(faces)
    for a, b in ((0, 1), (1, 0), (1, 2), (2, 1), (2, 0), (0, 2)):
        A[f[:, a], f[:, b]] = 1
    return A


def arap(dx, x, faces):
    """nnutils/loss_utils.py:46-64 (dense, for small V): [N]."""
    lap = torch.from_numpy(_adjacency(x.shape[1], faces)).to(x.dtype)
    diffx = torch.zeros(x.shape[0], x.shape[1], x.shape[1], dtype=x.dtype)
    diffdx = torch.zeros_like(diffx)
    for i in range(3):
        dx_sub = lap.matmul(torch.diag_embed(dx[:, :, i]))
        x_sub = lap.matmul(torch.diag_embed(x[:, :, i]))
        diffdx = diffdx + (dx_sub - dx[:, :, i:i + 1]).pow(2)
        diffx = diffx + (x_sub - x[:, :, i:i + 1]).pow(2)
    diff = (diffx - diffdx).abs()
    return torch.stack([diff[i][lap.bool()].mean() for i in range(x.shape[0])])


def laplacian(x, faces):
    """third_party/ext_nnutils/loss_utils.py:34-65 (dense): [N]."""
    nv = x.shape[1]
    L = -_adjacency(nv, faces)
    L[np.arange(nv), np.arange(nv)] = -L.sum(1)
    for i in range(nv):
        if L[i, i] != 0:
            L[i, :] /= L[i, i]
    y = torch.matmul(torch.from_numpy(L).to(x.dtype), x)
    return y.pow(2).sum((1, 2))


def flow_reproject(px, pp0, pp1, fl0, fl1):
    """nnutils/mesh_net.py:87-104 after the render: px [N,7,S,S] = (position at frame t, at frame t', alpha).
    -> flow [N,S,S,2], bgmask [N,S,S] bool.  Frame t's projection and background pixels are detached."""
    p0 = px[:, 0:3].permute(0, 2, 3, 1).clone()
    p1 = px[:, 3:6].permute(0, 2, 3, 1).clone()
    bg = (p0[..., 2] < 1e-9) | (p1[..., 2] < 1e-9)
    p0[bg] = 10
    p1[bg] = 10

    def proj(p, pp, fl):
        x = pp[:, 0:1, None] + p[..., 0] * fl[:, :1, None] / p[..., 2]
        y = pp[:, 1:2, None] + p[..., 1] * fl[:, :1, None] / p[..., 2]
        return torch.stack([x, y], -1)
    flow = proj(p1, pp1, fl1) - proj(p0, pp0, fl0).detach()
    flow = torch.where(bg[..., None], flow.detach(), flow)
    return flow, bg


def quaternion_to_rotation_matrix(q):
    """kornia 0.5.3 semantics (not vendored): (x,y,z,w), normalised first."""
    q = torch.nn.functional.normalize(q, p=2, dim=-1, eps=1e-12)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    m = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)
    return m.reshape(q.shape[:-1] + (3, 3))


def skin_weights(ctl_ts, ctl_rs, log_ctl, verts):
    """nnutils/mesh_net.py:264-271; ctl_* [H*J,.], verts [H,V,3] -> [H,J,V]."""
    H = verts.shape[0]
    dis = ctl_ts.view(H, -1, 1, 3) - verts[:, None].detach()
    dis = dis.matmul(quaternion_to_rotation_matrix(ctl_rs).view(H, -1, 3, 3))
    dis = log_ctl.exp().view(H, -1, 1, 3) * dis.pow(2)
    return (-10 * dis.sum(3)).softmax(1)


def point_mesh_face_distance(verts, faces, points):
    """pytorch3d.loss.point_mesh_face_distance semantics as used at nnutils/mesh_net.py:470-471 (pytorch3d 0.4.0 is not
    vendored: parity unpinned).  mean_n [ mean_p min_f d2(p, f) + mean_f min_p d2(p, f) ], brute force over P x F pairs;
    closest point by Voronoi region (Ericson, Real-Time Collision Detection 5.1.5) written as a min over candidates."""
    tri = verts[:, faces]                                                # [B,F,3,3]
    a, b, c = tri[:, None, :, 0], tri[:, None, :, 1], tri[:, None, :, 2]  # [B,1,F,3]
    p = points[:, :, None]                                                # [B,P,1,3]
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    va, vb, vc = d3 * d6 - d5 * d4, d5 * d2 - d1 * d6, d1 * d4 - d3 * d2
    eps = 1e-12
    denom = (va + vb + vc).clamp_min(eps)
    v, w = vb / denom, vc / denom
    inside = a + ab * v[..., None] + ac * w[..., None]
    t_ab = (d1 / (d1 - d3).clamp_min(eps)).clamp(0, 1)
    t_ac = (d2 / (d2 - d6).clamp_min(eps)).clamp(0, 1)
    t_bc = ((d4 - d3) / ((d4 - d3) + (d5 - d6)).clamp_min(eps)).clamp(0, 1)
    cands = torch.stack([inside, a + ab * t_ab[..., None], a + ac * t_ac[..., None], b + (c - b) * t_bc[..., None]], 0)
    ok_inside = (va >= 0) & (vb >= 0) & (vc >= 0)
    d = (cands - p).pow(2).sum(-1)                                        # [4,B,P,F]
    d_in = torch.where(ok_inside, d[0], torch.full_like(d[0], float('inf')))
    dist = torch.minimum(d_in, d[1:].min(0)[0])                           # [B,P,F]
    return (dist.min(2)[0].mean(1) + dist.min(1)[0].mean(1)).mean()


def chamfer_distance(a, b):
    """pytorch3d.loss.chamfer_distance()[0] semantics as used at nnutils/mesh_net.py:503 (parity unpinned)."""
    d = (a[:, :, None] - b[:, None]).pow(2).sum(-1)
    return (d.min(2)[0].mean(1) + d.min(1)[0].mean(1)).mean()


def load_textures(image, faces_uv, R, is_update=None):
    """third_party/softras/soft_renderer/cuda/load_textures_cuda_kernel.cu:8-66, numpy fp32 (bilinear branch).
    image [H,W,3], faces_uv [F,3,2] -> [F,R*R,3].  Indices that the reference reads one past the image (uv == 1, weight 0)
    are clamped."""
    image = np.asarray(image, np.float32)
    uv = np.asarray(faces_uv, np.float32)
    H, W = image.shape[:2]
    F = uv.shape[0]
    out = np.zeros((F, R * R, 3), np.float32)
    for i in range(R * R):
        w_y, w_x = i // R, i % R
        if w_x + w_y < R:
            w0, w1 = np.float32((w_x + 1. / 3.) / R), np.float32((w_y + 1. / 3.) / R)
        else:
            w0, w1 = np.float32(((R - 1. - w_x) + 2. / 3.) / R), np.float32(((R - 1. - w_y) + 2. / 3.) / R)
        w2 = np.float32(1. - np.float64(w0) - np.float64(w1))
        px = (uv[:, 0, 0] * w0 + uv[:, 1, 0] * w1 + uv[:, 2, 0] * w2) * np.float32(W - 1)
        py = (uv[:, 0, 1] * w0 + uv[:, 1, 1] * w1 + uv[:, 2, 1] * w2) * np.float32(H - 1)
        x0, y0 = px.astype(np.int32), py.astype(np.int32)
        wx1 = px - x0.astype(np.float32); wx0 = np.float32(1) - wx1
        wy1 = py - y0.astype(np.float32); wy0 = np.float32(1) - wy1
        xa, xb = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1)
        ya, yb = np.clip(y0, 0, H - 1), np.clip((py + np.float32(1)).astype(np.int32), 0, H - 1)
        c = np.zeros((F, 3), np.float32)
        c += image[ya, xa] * (wx0 * wy0)[:, None]
        c += image[yb, xa] * (wx0 * wy1)[:, None]
        c += image[ya, xb] * (wx1 * wy0)[:, None]
        c += image[yb, xb] * (wx1 * wy1)[:, None]
        out[:, i] = c
    if is_update is not None:
        out[np.asarray(is_update) == 0] = 0
    return out


def geodesic_distance(m1, m2):
    """third_party/ext_utils/util_rot.py:27-37: rotation angle between [n,3,3] batches (cos clamped to [-1, 1])."""
    m = torch.bmm(m1, m2.transpose(1, 2))
    cos = (m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2] - 1) / 2
    cos = torch.min(cos, torch.ones_like(cos))
    cos = torch.max(cos, -torch.ones_like(cos))
    return torch.acos(cos)


def intrinsics(cams, pp, scale, depth, ppoint, img_size):
    """nnutils/mesh_net.py:204-217, line by line: -> (scale [2B,H], depth [2B,K], ppoint [2B,2])."""
    B = cams.shape[0] // 2
    scale = cams[:, :1] * scale
    depth = torch.cat([cams[:, :1] * depth[:, :1], depth[:, 1:]], 1)
    ppb1 = cams[:B, :1] * pp[:B] / (img_size / 2.)
    ppb2 = cams[B:, :1] * pp[B:] / (img_size / 2.)
    ppa1 = ppoint[:B] + ppb1 + 1
    ppa2 = ppa1 * (cams[B:, :1] / cams[:B, :1])
    ppoint = torch.cat([ppoint[:B], ppa2 - ppb2 - 1], 0)
    return scale, depth, ppoint


def bone_fixup(quat, trans, depth, rest_ts, n_images, H, K):
    """nnutils/mesh_net.py:259-283, line by line: quat [M*K,9], trans [M*K,2], depth [M*K,1], rest_ts [H,(K-1)*3]
    (M = n_images * H) -> (Rmat [M*K,3,3], Tmat [M*K,3])."""
    Rmat = quat.view(-1, 3, 3).permute(0, 2, 1)
    Tmat = torch.cat([trans, depth], 1)
    if K > 1:
        rest = rest_ts[:, None, :, None].repeat(n_images, 1, 1, 1).view(-1, K - 1, 3, 1)
        Rmat = Rmat.reshape(-1, K, 3, 3)
        Tmat = Tmat.view(-1, K, 3, 1)
        Tmat = torch.cat([Tmat[:, :1], -Rmat[:, 1:].matmul(rest) + Tmat[:, 1:] + rest], 1)
        Rmat = torch.cat([Rmat[:, :1], Rmat[:, 1:].permute(0, 1, 3, 2)], 1)
    return Rmat.reshape(-1, 3, 3), Tmat.reshape(-1, 3)


def weighted_mean_sum(terms):
    """total_loss += w * x.mean(), term by term (nnutils/mesh_net.py:374-530): terms = [(tensor, weight, group)]
    -> (total, [group totals])."""
    total, groups = 0., {}
    for x, w, g in terms:
        v = w * x.mean()
        total = total + v
        groups[g] = groups.get(g, 0.) + v
    return total, [groups[g] for g in sorted(groups)]
