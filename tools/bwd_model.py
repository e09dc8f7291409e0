"""Survivors of the backward kernel's stage 1 per face on the bench frame (numpy model): bounding-box pixels, survivors of the
conservative line-distance reject, exact survivors, inside pixels; batches of 64 per face as the kernel runs them and if the ring were
carried across faces.   python tools/bwd_model.py"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from lasr_amd import synth
IS = 256
v, f, tex = synth.blobby_mesh(11)
tot = {}
for fr in (3, 40, 100, 200):
    pv = synth.frame_vertices(v, 256, first=fr, count=1)[0]
    tri = pv[f][:, :, :2]
    thr = 1e-4 * np.log(1. / 1e-4 - 1.)
    r = np.sqrt(thr); rp = np.sqrt(thr * 1.05)
    xs = (2 * np.arange(IS) + 1 - IS) / IS
    ys = (2 * (IS - 1 - np.arange(IS)) + 1 - IS) / IS
    S = []; B = []
    for t in tri:
        x0, x1 = t[:, 0].min() - r, t[:, 0].max() + r
        y0, y1 = t[:, 1].min() - r, t[:, 1].max() + r
        cols = np.where((xs >= x0) & (xs <= x1))[0]; rows = np.where((ys >= y0) & (ys <= y1))[0]
        B.append(len(cols) * len(rows))
        if len(cols) == 0 or len(rows) == 0: S.append(0); continue
        X, Y = np.meshgrid(xs[cols], ys[rows]); P = np.stack([X, Y], -1)
        # stage-1 reject: signed line distances
        keep = np.ones(X.shape, bool)
        area = (t[1,0]-t[0,0])*(t[2,1]-t[0,1]) - (t[1,1]-t[0,1])*(t[2,0]-t[0,0])
        sg = 1 if area > 0 else -1
        for a, b in ((0, 1), (1, 2), (2, 0)):
            A_, B_ = t[a], t[b]; ab = B_ - A_; n = np.hypot(*ab) + 1e-30
            d = sg * ((ab[0]) * (P[..., 1] - A_[1]) - (ab[1]) * (P[..., 0] - A_[0])) / n
            keep &= ~(d < -rp)
        S.append(int(keep.sum()))
    S = np.array(S); B = np.array(B)
    nb_now = np.ceil(S / 64).sum(); nb_dense = np.ceil(S.sum() / 64)
    s1 = np.ceil(B / 64).sum()
    print('frame', fr, 'faces', len(S), 'nonempty', (S > 0).sum(), 'bbox px', B.sum(), 'survivors', S.sum(), 'mean', S[S>0].mean(),
          'batches now', nb_now, 'dense', nb_dense, 'ratio', nb_dense / nb_now, 'stage1 rounds', s1, 'live s1', B.sum()/(64*s1))
    print('  hist survivors:', np.histogram(S, bins=[0,1,16,32,48,64,96,128,192,256,512,100000])[0])
print('--- exact survivors')
for fr in (3, 200):
    pv = synth.frame_vertices(v, 256, first=fr, count=1)[0]
    tri = pv[f][:, :, :2]
    S = []; I = []
    for t in tri:
        x0, x1 = t[:, 0].min() - r, t[:, 0].max() + r
        y0, y1 = t[:, 1].min() - r, t[:, 1].max() + r
        cols = np.where((xs >= x0) & (xs <= x1))[0]; rows = np.where((ys >= y0) & (ys <= y1))[0]
        if len(cols) == 0 or len(rows) == 0: S.append(0); I.append(0); continue
        X, Y = np.meshgrid(xs[cols], ys[rows]); P = np.stack([X, Y], -1)
        def cross(a, b, p): return (b[0] - a[0]) * (p[..., 1] - a[1]) - (b[1] - a[1]) * (p[..., 0] - a[0])
        c0, c1, c2 = cross(t[0], t[1], P), cross(t[1], t[2], P), cross(t[2], t[0], P)
        inside = ((c0 >= 0) & (c1 >= 0) & (c2 >= 0)) | ((c0 <= 0) & (c1 <= 0) & (c2 <= 0))
        d2 = np.full(X.shape, 1e9)
        for a, b in ((0, 1), (1, 2), (2, 0)):
            A_, B_ = t[a], t[b]; ab = B_ - A_; den = (ab * ab).sum() + 1e-30
            tt = np.clip(((P - A_) * ab).sum(-1) / den, 0, 1)
            q = A_ + tt[..., None] * ab
            d2 = np.minimum(d2, ((P - q) ** 2).sum(-1))
        keep = inside | (d2 < thr * 1.05)
        S.append(int(keep.sum())); I.append(int(inside.sum()))
    S = np.array(S); I = np.array(I)
    print('frame', fr, 'exact survivors', S.sum(), 'mean', S.mean(), 'inside', I.sum(), 'batches per-face', np.ceil(S/64).sum(), 'dense', np.ceil(S.sum()/64))
    print('  hist:', np.histogram(S, bins=[0,1,16,32,48,64,96,128,192,256,512,100000])[0])
