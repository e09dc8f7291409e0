#!/usr/bin/env python3
"""Why does the issue ORDER of the forward's 8x8 tiles matter, and is the shipped sort key good enough?  CPU model.

For `--frames` bench frames (mesh M2, 256x256, the bench's pose cycle) it computes per tile
  key    = faces whose padded pixel rect touches the tile (what sr_tile_weight_kernel counts, clamped at 255),
  heavy  = faces with at least one SURVIVING pixel in the tile (inside, or within the distance threshold): the entries that run
           the distance / sigmoid / depth code,
  cost   = 60 x key + 230 x heavy wave-instructions (walk + reject ~60 per entry, the rest of an entry ~230: DESIGN.md section 4),
then plays the launch on a model chip: 8 XCDs x 128 SIMDs, 8 wave slots per SIMD, workgroups handed out in issue order to the
SIMDs of the block's XCD (block b -> XCD b % 8) round robin (or to the one with the fewest resident waves), a SIMD sharing its issue slots among its resident
waves (a lone wave cannot use more than 1 / `--solo` of them: latency bound).  Orders compared: the fixed centre-out spiral
(rank-major over the XCD's images, as tile_of_block), descending `key` per XCD (shipped), descending `cost` per XCD (oracle).
Prints the makespan of each relative to the perfectly balanced bound, and the rank correlation of key and cost.

    python tools/tile_order_model.py [--frames 16] [--solo 3]       (CPU only, ~1 min for 16 frames)
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lasr_amd import synth          # noqa: E402

IS, T = 256, 32
ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=16)
ap.add_argument('--solo', type=float, default=3.0, help='a lone wave uses at most 1/solo of its SIMD (latency bound)')
ap.add_argument('--placement', choices=('rr', 'least'), default='rr', help='workgroup placement: round robin over the SIMDs with a free slot, or least loaded')
opts = ap.parse_args()
N = opts.frames
assert N % 8 == 0, 'model of the per-XCD sort: frames must be a multiple of 8'

v, f, tex = synth.blobby_mesh(11)
pv = synth.frame_vertices(v, 26, count=N)                      # [N,V,3]
thr = 1e-4 * np.log(1. / 1e-4 - 1.)
r = np.sqrt(thr)
xs = (2 * np.arange(IS) + 1 - IS) / IS
ys = (2 * (IS - 1 - np.arange(IS)) + 1 - IS) / IS              # row 0 = top


def frame_tables(tri):
    key = np.zeros((T, T), np.int64)
    heavy = np.zeros((T, T), np.int64)
    for t in tri:
        x0, x1 = t[:, 0].min() - r, t[:, 0].max() + r
        y0, y1 = t[:, 1].min() - r, t[:, 1].max() + r
        cols = np.where((xs >= x0) & (xs <= x1))[0]
        rows = np.where((ys >= y0) & (ys <= y1))[0]
        if len(cols) == 0 or len(rows) == 0:
            continue
        key[rows[0] >> 3:(rows[-1] >> 3) + 1, cols[0] >> 3:(cols[-1] >> 3) + 1] += 1
        X, Y = np.meshgrid(xs[cols], ys[rows])
        P = np.stack([X, Y], -1)

        def cross(a, b):
            return (b[0] - a[0]) * (P[..., 1] - a[1]) - (b[1] - a[1]) * (P[..., 0] - a[0])
        c0, c1, c2 = cross(t[0], t[1]), cross(t[1], t[2]), cross(t[2], t[0])
        inside = ((c0 >= 0) & (c1 >= 0) & (c2 >= 0)) | ((c0 <= 0) & (c1 <= 0) & (c2 <= 0))
        d2 = np.full(X.shape, 1e9)
        for a, b in ((0, 1), (1, 2), (2, 0)):
            A, B = t[a], t[b]
            ab = B - A
            tt = np.clip(((P - A) * ab).sum(-1) / ((ab * ab).sum() + 1e-30), 0, 1)
            q = A + tt[..., None] * ab
            d2 = np.minimum(d2, ((P - q) ** 2).sum(-1))
        rr, cc = np.where(inside | (d2 < thr))
        if len(rr):
            tiles = np.unique((rows[rr] >> 3) * T + (cols[cc] >> 3))
            heavy.reshape(-1)[tiles] += 1
    return key, heavy


keys, heavies = zip(*(frame_tables(pv[n][f][:, :, :2]) for n in range(N)))
key = np.minimum(np.stack(keys), 255)                          # [N,T,T]
cost = 60 * np.stack(keys) + 230 * np.stack(heavies) + 40     # + the tile's prologue and stores
print('frames %d: tiles %d, non-empty %.1f %%, key max %d mean %.1f; heavy entries / touched entries %.2f' %
      (N, key.size, 100 * (key > 0).mean(), key.max(), key.mean(), np.stack(heavies).sum() / max(np.stack(keys).sum(), 1)))
kf, cf = key.reshape(-1).astype(float), cost.reshape(-1).astype(float)
nz = kf > 0
rk = lambda a: np.argsort(np.argsort(a))                       # noqa: E731
print('rank correlation of key and cost over the non-empty tiles: %.4f' % np.corrcoef(rk(kf[nz]), rk(cf[nz]))[0, 1])


def spiral_cells():
    """rank -> (ty, tx) of tile_of_block's square spiral around the grid centre"""
    out = []
    h = T // 2
    for rank in range(T * T):
        rr = int(np.sqrt(rank) * 0.5)
        while 4 * rr * rr > rank:
            rr -= 1
        while 4 * (rr + 1) * (rr + 1) <= rank:
            rr += 1
        o, s = rank - 4 * rr * rr, 2 * rr + 1
        side, k = divmod(o, s)
        lo, hi = h - 1 - rr, h + rr
        out.append(((lo, lo + k), (lo + k, hi), (hi, hi - k), (hi - k, lo))[side])
    return out


def xcd_lists(order):
    """per XCD: list of tile costs in issue order"""
    m = N // 8
    lists = []
    for x in range(8):
        imgs = range(x * m, (x + 1) * m)
        if order == 'spiral':
            cells = spiral_cells()
            g = min(m, 4)
            lst = []
            for grp in range(0, m, g):
                for ty, tx in cells:
                    for n in imgs[grp:grp + g]:
                        lst.append(cost[n, ty, tx])
        else:
            which = key if order == 'key' else cost
            ids = [(n, ty, tx) for n in imgs for ty in range(T) for tx in range(T)]
            ids.sort(key=lambda e: -which[e])
            lst = [cost[e] for e in ids]
        lists.append(np.array(lst, float))
    return lists


def makespan(lists, solo):
    """event simulation of one XCD at a time (they are independent): 128 SIMDs x 8 slots, processor sharing"""
    worst = 0.0
    for lst in lists:
        S = 128
        resident = [[] for _ in range(S)]                      # remaining work of the waves on each SIMD
        nxt, t, cursor = 0, 0.0, 0
        while True:
            # hand out workgroups while a slot is free: to the SIMD with the fewest resident waves
            while nxt < len(lst):
                if opts.placement == 'least':                  # the SIMD with the fewest resident waves
                    s = min(range(S), key=lambda i: len(resident[i]))
                    if len(resident[s]) >= 8:
                        break
                else:                                          # round robin: the next SIMD in cyclic order that has a free slot
                    for k in range(S):
                        s = (cursor + k) % S
                        if len(resident[s]) < 8:
                            break
                    else:
                        break
                    cursor = s + 1
                resident[s].append(lst[nxt])
                nxt += 1
            busy = [i for i in range(S) if resident[i]]
            if not busy:
                break
            # rate of each wave on SIMD i: min(1 / n, 1 / solo) instructions per slot
            dt = min(min(resident[i]) / min(1.0 / len(resident[i]), 1.0 / solo) for i in busy)
            for i in busy:
                rate = min(1.0 / len(resident[i]), 1.0 / solo)
                resident[i] = [w - rate * dt for w in resident[i]]
                resident[i] = [w for w in resident[i] if w > 1e-9]
            t += dt
        worst = max(worst, t)
    return worst


ideal = cost.sum() / (8 * 128)
print('perfectly balanced bound: %.0f issue slots per SIMD' % ideal)
for order in ('spiral', 'key', 'cost'):
    ms = makespan(xcd_lists(order), opts.solo)
    print('%-7s makespan %.0f  = %.3f x the bound' % (order, ms, ms / ideal))
