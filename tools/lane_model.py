#!/usr/bin/env python3
"""How many lanes of a wave can the forward walk keep busy?  CPU model of one bench frame (mesh M2, 256x256): the surviving
(pixel, face) pairs of the reference's distance threshold are enumerated exactly, then binned into pixel blocks with their own
face lists; a wave covers one or several blocks and iterates max(list length) times.  Prints wave-iterations and the live-lane
fraction for the shipped 8x8 quadrant and for split-wave alternatives (profiles/experiments/README.md)."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from lasr_amd import synth
IS = 256
v, f, tex = synth.blobby_mesh(11)
pv = synth.frame_vertices(v, 26, first=3, count=1)[0]          # [V,3] one frame (NDC x,y)
tri = pv[f][:, :, :2]                                         # [F,3,2]
thr = 1e-4 * np.log(1. / 1e-4 - 1.)
r = np.sqrt(thr)
xs = (2 * np.arange(IS) + 1 - IS) / IS
ys = (2 * (IS - 1 - np.arange(IS)) + 1 - IS) / IS            # row 0 = top
surv = []                                                     # per face: (rows, cols) of surviving pixels
for t in tri:
    x0, x1 = t[:, 0].min() - r, t[:, 0].max() + r
    y0, y1 = t[:, 1].min() - r, t[:, 1].max() + r
    cols = np.where((xs >= x0) & (xs <= x1))[0]; rows = np.where((ys >= y0) & (ys <= y1))[0]
    if len(cols) == 0 or len(rows) == 0: surv.append((np.zeros(0, int), np.zeros(0, int))); continue
    X, Y = np.meshgrid(xs[cols], ys[rows])
    P = np.stack([X, Y], -1)                                  # [R,C,2]
    # inside test
    def cross(a, b, p): return (b[0] - a[0]) * (p[..., 1] - a[1]) - (b[1] - a[1]) * (p[..., 0] - a[0])
    c0, c1, c2 = cross(t[0], t[1], P), cross(t[1], t[2], P), cross(t[2], t[0], P)
    inside = ((c0 >= 0) & (c1 >= 0) & (c2 >= 0)) | ((c0 <= 0) & (c1 <= 0) & (c2 <= 0))
    d2 = np.full(X.shape, 1e9)
    for a, b in ((0, 1), (1, 2), (2, 0)):
        A, B = t[a], t[b]; ab = B - A; den = (ab * ab).sum() + 1e-30
        tt = np.clip(((P - A) * ab).sum(-1) / den, 0, 1)
        q = A + tt[..., None] * ab
        d2 = np.minimum(d2, ((P - q) ** 2).sum(-1))
    keep = inside | (d2 < thr)
    rr, cc = np.where(keep)
    surv.append((rows[rr], cols[cc]))
pairs = sum(len(a) for a, _ in surv)
print('faces', len(f), 'surviving pairs per frame', pairs)
def count(bh, bw, group_h, group_w):
    """blocks of bh x bw pixels, each with its own face list; a wave = group_h x group_w blocks (= 64 lanes); iterations of a
    wave = max list length over its blocks.  Returns (wave iterations, live-lane fraction)."""
    nby, nbx = IS // bh, IS // bw
    L = np.zeros((nby, nbx), int)
    for rr, cc in surv:
        if len(rr) == 0: continue
        b = np.unique((rr // bh) * nbx + (cc // bw))
        L.reshape(-1)[b] += 1
    W = L.reshape(nby // group_h, group_h, nbx // group_w, group_w).max(axis=(1, 3))
    it = W.sum()
    return int(it), pairs / (it * 64.0), int(L.sum())
for name, args in (('8x8 quadrant per wave (now)', (8, 8, 1, 1)), ('two 4x8 halves per wave', (4, 8, 2, 1)), ('two 8x4 halves per wave', (8, 4, 1, 2)),
                   ('four 4x4 blocks per wave', (4, 4, 2, 2)), ('16x4 strip per wave', (4, 16, 1, 1)), ('eight 2x4 blocks', (2, 4, 4, 2))):
    it, lanes, entries = count(*args)
    print('%-32s wave-iterations %6d  live lanes %.3f  block entries %d' % (name, it, lanes, entries))


def chunked(chunk):
    """Four 4x4 blocks per wave walking a SHARED staged window: the tile's entry list (index order) is cut into chunks of `chunk`
    entries, the wave iterates max over its four blocks of the block's entries inside the chunk (the measured variant,
    profiles/experiments/README.md round 3)."""
    nb = IS // 4
    per_block = [[] for _ in range(nb * nb)]                  # face ids per 4x4 block, index order
    for fi, (rr, cc) in enumerate(surv):
        if len(rr) == 0: continue
        for b in np.unique((rr // 4) * nb + (cc // 4)):
            per_block[b].append(fi)
    total = 0
    for ty in range(IS // 8):
        for tx in range(IS // 8):
            blocks = [per_block[(2 * ty + dy) * nb + 2 * tx + dx] for dy in (0, 1) for dx in (0, 1)]
            tile = sorted(set().union(*blocks))
            if not tile: continue
            pos = {fid: i for i, fid in enumerate(tile)}
            n_chunks = (len(tile) + chunk - 1) // chunk
            cnt = np.zeros((4, n_chunks), int)
            for b, lst in enumerate(blocks):
                for fid in lst: cnt[b, pos[fid] // chunk] += 1
            total += int(cnt.max(axis=0).sum())
    return total


for c in (16, 32, 64, 1 << 20):
    print('four 4x4 blocks, shared window of %7d entries: wave-iterations %d' % (c, chunked(c)))


def longest(bh, bw):
    nby, nbx = IS // bh, IS // bw
    L = np.zeros(nby * nbx, int)
    for rr, cc in surv:
        if len(rr): L[np.unique((rr // bh) * nbx + (cc // bw))] += 1
    return int(L.max()), float(L[L > 0].mean())


# the serial walk of the busiest block sets the floor of a small launch (one frame: 0.21 ms)
for bh, bw in ((8, 8), (4, 8), (4, 4), (2, 4), (2, 2)):
    print('longest list of a %dx%d block: %d entries (mean over non-empty blocks %.1f)' % ((bh, bw) + longest(bh, bw)))
