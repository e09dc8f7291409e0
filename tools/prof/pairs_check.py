#!/usr/bin/env python3
"""Pair-walk forward kernel (lasr_amd/csrc/sr_forward_pairs.h) against the one-wave-per-tile kernel on an MI355X:
   python tools/prof/pairs_check.py parity        image / aggregate differences on ragged sizes, 3 / 6 / 9 channels, long lists
   python tools/prof/pairs_check.py time [N ...]  per-kernel times of the bench step with either kernel forced (child processes)
"""
import os, sys, json, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
BIG = 10 ** 12


def parity():
    import torch
    from lasr_amd import synth
    from lasr_amd.soft_renderer import functional as srf
    dev = torch.device('cuda:0')

    def render(fv, ft, IS, kw, pairs):
        srf.set_launch_thresholds(0, 0, 0, -1, 0 if pairs else BIG)
        img = srf.soft_rasterize(torch.from_numpy(fv).to(dev), torch.from_numpy(ft).to(dev), IS, **kw)
        torch.cuda.synchronize()
        return img.cpu().numpy()
    worst = 0.
    cases = []
    for IS in (1, 7, 8, 20, 33, 64, 100, 256):
        fv, ft, near, far = synth.raster_batch(4, 3, count=3)
        cases.append(('ragged IS=%d' % IS, fv, ft, IS, dict(synth.LASR_MODES, near=near, far=far)))
    for ch in (6, 9):
        fv, ft, near, far = synth.raster_batch(4, 3, count=2)
        rng = np.random.default_rng(ch)
        tex = np.concatenate([ft] + [rng.uniform(-2, 2, ft.shape).astype(np.float32) for _ in range(ch // 3 - 1)], -1)
        cases.append(('%d channels' % ch, fv, tex, 72, dict(synth.LASR_MODES, near=near, far=far, background_color=[0.25 * k for k in range(ch)])))
    rng = np.random.default_rng(7)
    F = 2500
    c = rng.uniform(-0.15, 0.15, (2, F, 1, 2)); tri = c + rng.uniform(-0.08, 0.08, (2, F, 3, 2)); z = rng.uniform(2, 4, (2, F, 3, 1))
    fv = np.concatenate([tri, z], -1).astype(np.float32)
    fv[0, 0] = [[-1.5, -1.2, 3], [1.4, -1.1, 3.5], [0.1, 1.6, 2.5]]
    cases.append(('2500 faces in the centre', fv, rng.uniform(0, 1, fv.shape).astype(np.float32), 64, dict(synth.LASR_MODES, near=1.0, far=5.0)))
    for nu, count in ((8, 16), (11, 4), (11, 64)):
        fv, ft, near, far = synth.raster_batch(nu, 3, count=count)
        cases.append(('nu=%d x %d @256' % (nu, count), fv, ft, 256, dict(synth.LASR_MODES, near=near, far=far)))
    # degenerate / non-tame faces: a sliver, a zero-area face, a face behind the near plane
    fv, ft, near, far = synth.raster_batch(4, 3, count=2)
    fv = fv.copy()
    fv[0, 0] = [[0, 0, 3], [0.5, 0.5, 3], [1e-7, 0, 3]]
    fv[0, 1] = [[0.1, 0.1, 3], [0.1, 0.1, 3], [0.1, 0.1, 3]]
    fv[1, 2, :, 2] = 1e-9
    cases.append(('degenerate faces', fv, ft, 64, dict(synth.LASR_MODES, near=near, far=far)))
    for name, fv, ft, IS, kw in cases:
        a = render(fv, ft, IS, kw, False)
        b = render(fv, ft, IS, kw, True)
        d = float(np.abs(a - b).max())
        nan = int(np.isnan(b).sum())
        worst = max(worst, d)
        print('%-28s shape %-20s max |one-wave - pairs| %.3e  nan %d  (max |img| %.3f)' % (name, a.shape, d, nan, np.abs(a).max()), flush=True)
    srf.set_launch_thresholds()
    print('worst', worst)


def time_child(n, steps):
    import torch, ctypes
    import bench
    from lasr_amd import _lib
    dev = torch.device('cuda:0')
    st = bench.RasterStep(dev, n, 0)
    h = _lib.lib()
    for _ in range(3): st.step()
    torch.cuda.synchronize()
    h.lasr_prof_enable(st.stream, 1)
    for _ in range(steps): st.step()
    torch.cuda.synchronize()
    t = bench.collect_kernel_times(h, st.stream)
    h.lasr_prof_enable(st.stream, 0)
    print('RESULT ' + json.dumps({k: v[0] for k, v in t.items()}))


def time_(ns):
    for n in ns:
        row = {}
        for name, env in (('one-wave', str(BIG)), ('pairs', '0')):
            e = dict(os.environ, LASR_SR_PAIR_MIN_TILES=env)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), 'time-child', str(n), str(max(3, 600 // n))], env=e, capture_output=True, text=True)
            r = [l for l in out.stdout.splitlines() if l.startswith('RESULT ')]
            if not r:
                print(out.stdout[-2000:], out.stderr[-2000:]); continue
            row[name] = json.loads(r[0][7:])
        print('frames %4d' % n, ' | '.join('%s: %s' % (k, ' '.join('%s %.4f' % (kk.replace('sr_', ''), vv) for kk, vv in v.items() if 'sr_' in kk)) for k, v in row.items()), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'parity': parity()
    elif sys.argv[1] == 'time-child': time_child(int(sys.argv[2]), int(sys.argv[3]))
    else: time_([int(a) for a in sys.argv[2:]] or [256, 64, 16, 4])
