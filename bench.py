#!/usr/bin/env python3
"""bench.py -- soft-rasteriser forward+backward throughput on MI355X (BASELINE.json metric #1).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames B]      # N > 1: re-launches itself, one process per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          # the same thing, launched explicitly
(the reference launches one process per GPU from its script line as well: scripts/template.sh:26, optimize.py:42-47)

Workload (BASELINE configs[1]): mesh M2 (geodesic nu=11, V=1212, F=2420,
the "~1.2k vert / 2.3k face" mesh), 256x256, LASR's raster modes (euclidean / softmax /
prod / vertex colours, sigma 1e-4, gamma 1e-2), B=256 synthetic yaw-rotated frames per GPU per
step, upstream gradient N(0,1)/P.  Inputs are resident in HBM before the timed region.

One step = for the rank's B frames: forward (face setup + raster kernel; the background colour is an argument, every element of
soft_colors is written), backward (face setup + raster kernel; it stores every gradient element) through the C ABI,
reduce the face gradients to per-vertex gradients (the product's deterministic lasr_face_gather_backward, CSR form) and sum them over the
frames; for N > 1 the [2,V,3] mesh
gradient is then all-reduced over RCCL (frames are sharded data-parallel, weak scaling).
Nothing inside the timed region touches the CPU oracle.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP-event timed inside the
library on the launch stream) and, at N=1, `cpu_baseline` (the oracle, OpenMP over host cores,
on a bounded sample of the same workload).
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from lasr_amd import _lib, parallel, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3       # dense fp32-input MFMA peak (same guide)
IS = 256
NU = 11                        # mesh M2
N_FRAMES_CYCLE = 26            # yaw positions ("~26 frames" of BASELINE configs)
REBUILD_RECORDS = -1            # -1: what the autograd operator does (reuse the forward's records)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--frames', type=int, default=256, help='frames per GPU per step (SURVEY 8d batches: 1/16/64/256)')
    ap.add_argument('--image-size', type=int, default=256, help='256 = the headline metric; 512 = BASELINE configs[2] (camel)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--rebuild-records', type=int, default=-1, help='1: the backward rebuilds the per-face records, 0: it reuses the '
                    'forward\'s (LASR_SR_RECORDS_VALID), -1 (default): as the autograd operator does -- reuse')
    ap.add_argument('--no-lbs', action='store_true', help='skip the LBS (MFMA) micro-benchmark block')
    ap.add_argument('--no-sweep', action='store_true', help='skip the launch-size sweep (N = 1/4/16/64 at 256^2, 64 at 512^2)')
    ap.add_argument('--lasr-iters', type=int, default=20, help='optimize.py-style iterations timed at N=1 (0 = skip)')
    ap.add_argument('--no-step-profile', action='store_true', help='skip the in_scope_step block (a rocprofv3 kernel trace of one '
                    'optimisation iteration per configuration, taken in a child process)')
    ap.add_argument('--step-worker', default='', help=argparse.SUPPRESS)       # child mode of the in_scope_step block
    return ap.parse_args()


class RasterStep:
    """Pre-allocated buffers + one fwd/bwd pass through the C ABI."""

    def __init__(self, dev, B, first_frame, image_size=None, sigma=None, n_cycle=None):
        global IS
        IS = image_size or IS
        self.IS = IS
        self.dev, self.B = dev, B
        v, f, tex = synth.blobby_mesh(NU)
        self.V, self.F = v.shape[0], f.shape[0]
        n_cycle = n_cycle or N_FRAMES_CYCLE
        pv = synth.frame_vertices(v, n_cycle, first=first_frame, count=B)
        self.near, self.far = synth.near_far(synth.frame_vertices(v, n_cycle)[:, :, 2])
        self.faces_idx = torch.from_numpy(f).to(dev)
        self.fv = torch.from_numpy(np.ascontiguousarray(pv[:, f])).to(dev).reshape(B, self.F, 9).contiguous()
        self.ft = torch.from_numpy(np.ascontiguousarray(tex[f])).to(dev).reshape(1, self.F, 9).repeat(B, 1, 1).contiguous()
        self.g = torch.from_numpy(synth.upstream_grad(B, IS)).to(dev)
        self.colors = torch.empty(B, 4, IS, IS, device=dev)
        self.aggrs = torch.empty(B, 2, IS, IS, device=dev)
        self.gf = torch.empty(B, self.F, 9, device=dev)
        self.gt = torch.empty(B, self.F, 9, device=dev)
        self.mesh_grad = torch.zeros(2, self.V, 3, device=dev)    # d/d(vertex xyz), d/d(vertex colour), summed over frames
        self.scatter_idx = self.faces_idx.reshape(-1)             # [F*3]
        m = synth.LASR_MODES
        self.h = _lib.lib()
        self.ws = torch.empty(self.h.lasr_sr_workspace_bytes(B, self.F, 3, IS), dtype=torch.uint8, device=dev)
        self.scalars = (float(self.near), float(self.far), float(m['eps']), float(sigma if sigma is not None else m['sigma_val']), 2,
                        float(math.log(1. / m['dist_eps'] - 1.)), float(m['gamma_val']), 1, 2, 1, 1)
        self.stream = torch.cuda.current_stream(dev).cuda_stream
        self.white = (ctypes.c_float * 3)(1., 1., 1.)
        self.forward_flags = 0                                    # per-call flag of the forward pass (_lib.SR_RELAXED_MATH: opt-in)
        self.options = None                                       # per-call lasr_sr_options (None = the library's defaults)
        # face -> vertex reduction of the product (lasr_face_gather_backward_csr, geometry.py: _FaceGather): per-frame vertex gradients
        self.faces_n = self.faces_idx[None].expand(B, self.F, 3).contiguous()
        self.gv = torch.empty(2, B, self.V, 3, device=dev)
        # (the connectivity is fixed: its vertex -> corner incidence is built once, as the operator does for a face tensor it
        # sees again -- soft_renderer/functional/geometry.py: _incidence_of)
        from lasr_amd.nnutils import fused_ops
        self.inc_ptr, self.inc = fused_ops.face_incidence(self.faces_idx[None], self.V)

    def step(self):
        B, F, h, IS = self.B, self.F, self.h, self.IS
        # No background fill and no gradient zeroing passes: the background colour (1,1,1) is an argument of lasr_sr_forward_bg,
        # which writes every element of soft_colors (the reference pre-fills and re-reads it, soft_rasterize.py:50-53), and the
        # backward stores every gradient element (LASR_SR_GRADS_OVERWRITE).
        near, far, tail = self.scalars[0], self.scalars[1], self.scalars[2:]
        rc = h.lasr_sr_forward_opt(self.fv.data_ptr(), self.ft.data_ptr(), None, self.aggrs.data_ptr(),
                                   self.colors.data_ptr(), self.ws.data_ptr(), self.ws.numel(),
                                   B, F, 3, 3, IS, near, far, None, *tail, self.white, self.forward_flags,
                                   ctypes.byref(self.options) if self.options is not None else None, self.stream)
        _lib.check(rc, 'lasr_sr_forward_opt')
        # the per-face records in the backward, as the autograd operator handles them (soft_rasterize.py: _records_of): the
        # forward's are reused, one launch less (profiles/r04_flag_sweep.txt; --rebuild-records 1 times the other way)
        rebuild = REBUILD_RECORDS > 0
        rc = h.lasr_sr_backward_ex(self.fv.data_ptr(), self.ft.data_ptr(), self.colors.data_ptr(), self.aggrs.data_ptr(),
                                   self.gf.data_ptr(), self.gt.data_ptr(), self.g.data_ptr(), self.ws.data_ptr(),
                                   self.ws.numel(), B, F, 3, 3, IS, near, far, None, *tail,
                                   _lib.SR_GRADS_OVERWRITE | (0 if rebuild else _lib.SR_RECORDS_VALID), self.stream)
        _lib.check(rc, 'lasr_sr_backward_ex')
        # face -> vertex gradient reduction (autograd of face_vertices.py:4-22) with the product's own kernel -- deterministic:
        # a vertex sums its incident corners in ascending order, no atomics (the reference's autograd uses atomic index_add_) --
        # then the sum over the rank's frames (the mesh is shared by the frames: nnutils/mesh_net.py:255-283)
        for k, gsrc in enumerate((self.gf, self.gt)):
            _lib.check(h.lasr_face_gather_backward_csr(gsrc.data_ptr(), self.inc_ptr.data_ptr(), self.inc.data_ptr(), 1,
                                                       self.gv[k].data_ptr(), B, self.V, F, 3, self.stream), 'lasr_face_gather_backward_csr')
        torch.sum(self.gv, dim=1, out=self.mesh_grad)
        return self.mesh_grad


def collect_kernel_times(h, stream):
    out = {}
    for k in range(h.lasr_prof_kernel_count()):
        ms, n = ctypes.c_double(0), ctypes.c_longlong(0)
        h.lasr_prof_collect(stream, k, ctypes.byref(ms), ctypes.byref(n))
        if n.value:
            out[h.lasr_prof_kernel_name(k).decode()] = (ms.value / n.value, n.value)
    return out


def host_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container on a 256-thread
    host is often limited to a fraction of it; os.cpu_count() does not see that)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota not in ('max', '-1'):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(F):
    """Oracle (CPU restatement of the reference algorithm) on a bounded sample of the same workload.  The OpenMP thread count
    is calibrated first (one short pass per candidate up to the usable CPUs; oversubscribing a quota-limited container is
    several times slower than matching it), then one warm-up pass and the median of 5 timed fwd+bwd passes (forward parallel
    over pixels, backward over (image, row band) tasks with per-band gradient slabs folded in band order), and the median of
    3 passes of ONE frame on one thread."""
    from oracle import sr_oracle
    usable = host_cpus()
    sr_oracle.lib()
    kw = None

    def one_pass(fv_, ft_, g_, nb):
        t0 = time.perf_counter()
        ref = sr_oracle.forward(fv_, ft_, IS, **kw)
        sr_oracle.backward_banded(ref, g_, nb, IS, **kw)
        return time.perf_counter() - t0

    def sample(n):
        fv, ft, near, far = synth.raster_batch(NU, N_FRAMES_CYCLE, count=n)
        return fv, ft, synth.upstream_grad(n, IS), dict(synth.LASR_MODES, near=near, far=far)

    # ---- calibration: 4 frames per candidate thread count
    fv, ft, g, kw = sample(4)
    cands = sorted({t for t in (4, 8, 16, 32, 64, 128, 256, usable) if t <= usable}) or [1]
    calib = {}
    for t in cands:
        sr_oracle.set_threads(t)
        one_pass(fv, ft, g, max(1, min(IS // 8, -(-2 * t // 4))))
        calib[t] = 4 / one_pass(fv, ft, g, max(1, min(IS // 8, -(-2 * t // 4))))
        if len(calib) > 1 and calib[t] < 0.8 * max(calib.values()):
            break                                   # past the knee: more threads only get slower
    threads = max(calib, key=calib.get)
    # ---- timed passes
    n = max(4, min(2 * threads, 64))                # bounded sample: a few seconds per pass at most
    bands = max(1, min(IS // 8, -(-2 * threads // n)))
    fv, ft, g, kw = sample(n)
    threads = sr_oracle.set_threads(threads)
    one_pass(fv, ft, g, bands)                      # warm-up (page faults, OpenMP pool)
    t_all = sorted(one_pass(fv, ft, g, bands) for _ in range(5))
    sr_oracle.set_threads(1)
    one_pass(fv[:1], ft[:1], g[:1], 1)
    t_one = sorted(one_pass(fv[:1], ft[:1], g[:1], 1) for _ in range(3))
    sr_oracle.set_threads(threads)
    return {'value': n / t_all[2], 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'host_logical_cpus': os.cpu_count(), 'usable_cpus': usable,
            'threads_busy_backward': min(threads, n * bands), 'one_thread_frames_per_s': 1.0 / t_one[1],
            'thread_calibration_frames_per_s': {str(k): round(v, 2) for k, v in calib.items()},
            'passes_s': [round(t, 4) for t in t_all],
            'sample': '%d frames fwd+bwd of the same M2 %dx%d workload: 1 warm-up + median of 5 passes with %d OpenMP threads '
                      '(the fastest of the calibrated counts; backward: %d image x row-band tasks); 1-thread figure: median of '
                      '3 passes of 1 frame' % (n, IS, IS, threads, n * bands)}


LBS_SIZES = {'S0': (16, 642, 21), 'dog15': (6, 1282, 36), 'batch256': (256, 1212, 36)}


def _lbs_case(dev, N, V, K):
    gen = torch.Generator(device='cpu').manual_seed(0)
    v = torch.randn(N, V, 3, generator=gen).to(dev).requires_grad_(True)
    R = torch.randn(N * K, 3, 3, generator=gen).to(dev).requires_grad_(True)
    T = torch.randn(N * K, 1, 3, generator=gen).to(dev).requires_grad_(True)
    sk = torch.softmax(torch.randn(N, K - 1, V, 1, generator=gen), 1).to(dev).requires_grad_(True)
    g = torch.randn(N, V, 3, generator=gen).to(dev)
    return v, R, T, sk, g


def lbs_worker():
    """Child process of lbs_leg (run under rocprofv3 --kernel-trace).  Per size, in LBS_SIZES order: LBS_EAGER_CALLS eager forward +
    backward calls, then one forward + backward captured as a HIP graph and replayed LBS_TRACE_REPS times -- the regime the
    optimisation step runs these kernels in (eager launches of kernels this short are timed with whatever the previous launch
    left in flight: 5.4 and 9.0 us for the same S0 forward on two boxes)."""
    from lasr_amd.nnutils import geom_utils
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    for name, (N, V, K) in LBS_SIZES.items():
        v, R, T, sk, g = _lbs_case(dev, N, V, K)
        ins = (v, R, T, sk)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(LBS_EAGER_CALLS):
                torch.autograd.grad(geom_utils.obj_to_cam(v, R, T, K, 1, sk), ins, g)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            grads = torch.autograd.grad(geom_utils.obj_to_cam(v, R, T, K, 1, sk), ins, g)
        for _ in range(LBS_TRACE_REPS):
            graph.replay()
        torch.cuda.synchronize()
        del graph, grads
    print('lbs-worker done', flush=True)


LBS_TRACE_REPS = 40            # graph replays per size (the trace keeps the last 30)
LBS_EAGER_CALLS = 10           # eager calls per size before the capture (dropped by lbs_trace)


def under_profiler():
    """True when this process is itself running under rocprofv3 (its tool library is preloaded): the legs that start a rocprofv3 child
    of their own are skipped then -- a profiler inside a profiled process, possibly one collecting PMC counters, is not worth a hung
    or crashed box for two extra blocks of the line."""
    return bool(os.environ.get('ROCP_TOOL_LIBRARIES')) or 'rocprofiler' in os.environ.get('LD_PRELOAD', '') or \
        os.environ.get('LASR_BENCH_UNDER_PROFILER') == '1'                # (the last: the test of this switch)


def lbs_trace():
    """Kernel durations of the LBS launches from a rocprofv3 kernel trace of lbs_worker (dispatch end - start; the last 30 of the
    40 graph replays per size).  None when the profiler is not available."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None
    if under_profiler():
        return {'error': 'skipped: bench.py is itself running under rocprofv3'}
    tmp = tempfile.mkdtemp(prefix='lasr_lbs_', dir='/tmp')
    try:
        cmd = ['rocprofv3', '--kernel-trace', '-d', tmp, '-o', 't', '--', sys.executable, os.path.abspath(__file__), '--step-worker', 'lbs']
        r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        dbs = glob.glob(os.path.join(tmp, '**', '*.db'), recursive=True)
        if r.returncode != 0 or not dbs:
            return {'error': 'rc %d: %s' % (r.returncode, r.stdout.decode(errors='replace')[-300:])}
        c = sqlite3.connect(dbs[0])
        rows = list(c.execute('select s.kernel_name, d.end - d.start from rocpd_kernel_dispatch d '
                              'join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start'))
        out = {}
        for kern in ('lbs_forward_kernel', 'lbs_backward_mfma_kernel', 'lbs_backward_fold_kernel'):
            d = [ns / 1e3 for nm, ns in rows if kern in nm]
            per = LBS_EAGER_CALLS + LBS_TRACE_REPS
            if len(d) != per * len(LBS_SIZES):
                out[kern] = {'error': '%d dispatches, expected %d' % (len(d), per * len(LBS_SIZES))}
                continue
            for i, name in enumerate(LBS_SIZES):
                chunk = d[i * per + LBS_EAGER_CALLS + 10:(i + 1) * per]
                out.setdefault(name, {})[kern] = round(sum(chunk) / len(chunk), 2)
        return out
    except Exception as e:
        return {'error': repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def lbs_leg(dev):
    """The one MFMA user on the path (north_star): linear-blend skinning, forward `skin^T[V,K-1] x RT[K-1,12]` and, since round 5, the
    backward's three contractions (blended transform, g_skin = G x RT^T, g_RT = skin x G) on v_mfma_f32_16x16x4_f32.  us per
    launch at the S0 / dog15 sizes of SURVEY section 8 and a large batch, two clocks: the library's own HIP events around each
    launch of eager calls (a few us of event overhead on kernels this short), and the dispatch timestamps of a rocprofv3 kernel
    trace of the same forward + backward replayed as a HIP graph in a child process (`trace_us`, the figure that compares with
    the step profile).
    Forward flops = 2 N V (K-1) 12 + 2 N V 12; backward = 3 contractions of 2 N V (K-1) 12 (+ the per-vertex products)."""
    from lasr_amd.nnutils import geom_utils
    h = _lib.lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    out = {'mfma_instruction': 'v_mfma_f32_16x16x4_f32', 'peak_tflops': MFMA_F32_PEAK_TF, 'sizes': {},
           'clock': 'library HIP events around each launch (lasr_prof_*), mean of 100 eager calls; trace_us: rocprofv3 kernel trace of 30 HIP-graph replays of one forward + backward'}
    trace = lbs_trace()
    for name, (N, V, K) in LBS_SIZES.items():
        v, R, T, sk, g = _lbs_case(dev, N, V, K)
        for _ in range(5):
            geom_utils.obj_to_cam(v, R, T, K, 1, sk).backward(g)
        torch.cuda.synchronize()
        reps = 100
        h.lasr_prof_enable(st, 1)
        for _ in range(reps):
            geom_utils.obj_to_cam(v, R, T, K, 1, sk).backward(g)
        torch.cuda.synchronize()
        h.lasr_prof_enable(st, 0)
        kt = collect_kernel_times(h, st)
        us_f = kt['lbs_forward_kernel'][0] * 1e3
        us_b = kt['lbs_backward_kernel'][0] * 1e3
        us_fold = kt.get('lbs_backward_fold_kernel', (0.0, 0))[0] * 1e3
        flops_f = 2 * N * V * (K - 1) * 12 + 2 * N * V * 12
        flops_b = 3 * 2 * N * V * (K - 1) * 12 + 2 * N * V * 30
        mfma_f = N * -(-V // 16) * -(-(K - 1) // 4)        # one 16x16x4 instruction per 16 vertices x 4 bones (12 of 16 columns used)
        mfma_b = N * -(-V // 16) * (-(-(K - 1) // 4) + -(-(K - 1) // 16) * (3 + 4))
        tr = (trace or {}).get(name, {}) if isinstance(trace, dict) else {}
        out['sizes'][name] = {'N': N, 'V': V, 'K': K, 'us_per_call': round(us_f, 2), 'backward_us_per_call': round(us_b, 2),
                              'backward_fold_us_per_call': round(us_fold, 2), 'mfma_instructions': mfma_f,
                              'backward_mfma_instructions': mfma_b,
                              'trace_us': {'forward': tr.get('lbs_forward_kernel'), 'backward': tr.get('lbs_backward_mfma_kernel'),
                                           'backward_fold': tr.get('lbs_backward_fold_kernel')},
                              'achieved_tflops': flops_f / us_f / 1e6, 'frac_of_mfma_peak': flops_f / us_f / 1e6 / MFMA_F32_PEAK_TF,
                              'backward_achieved_tflops': flops_b / us_b / 1e6,
                              'backward_frac_of_mfma_peak': flops_b / us_b / 1e6 / MFMA_F32_PEAK_TF,
                              'achieved_GBs': N * (24 * V + 4 * V * (K - 1) + 48 * K) / us_f / 1e3,
                              'backward_achieved_GBs': N * (36 * V + 8 * V * (K - 1) + 96 * K) / us_b / 1e3}
    if isinstance(trace, dict) and 'error' in trace:
        out['trace_error'] = trace['error']
    return out


def optimize_leg(dev, iters):
    """BASELINE.json's second figure: optimize.py iterations/s on the spot3 stage-0 configuration
    (scripts/spot3.sh:24: B=1 pair, 8 hypotheses, 21 bones, icosphere-3, 256x256), synthetic 3-frame sequence,
    full step = encoder + LBS + the render (one nine-attribute pass for the reference's three calls) fwd/bwd + loss tables +
    regularisers + clipping / NaN guard / AdamW."""
    import optimize
    from lasr_amd.nnutils import train_utils
    opts = optimize.parse_flags(['--name', 'bench', '--checkpoint_dir', '', '--only_mean_sym', '--nouse_gtpose',
                                 '--subdivide', '3', '--n_bones', '21', '--n_hypo', '8', '--num_epochs', '5',
                                 '--batch_size', '1', '--opt_tex', 'yes', '--iters_per_epoch', str(iters + 6), '--use_graph'])
    opts.local_rank = dev.index
    torch.manual_seed(0)
    tr = train_utils.LASRTrainer(opts).init_training()
    tr.model.train()
    tr.reinit_bones()
    for i in range(6):                               # iteration 0 renders the part image; 1.. capture + replay the graph
        tr.module.iters = i
        tr.train_step(tr.set_input(tr.dataloader[i]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        tr.module.iters = 6 + i
        loss, _ = tr.train_step(tr.set_input(tr.dataloader[6 + i]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {'iters_per_s': iters / dt, 'ms_per_iter': dt / iters * 1e3, 'iters': iters, 'final_loss': float(loss),
            'config': 'spot3 stage 0: batch 1 pair, n_hypo 8, n_bones 21, V=642/F=1280, 256x256; per iteration ONE nine-attribute '
                      'render of the 16 (image, hypothesis) meshes fwd+bwd (texture colours + both flow position triples; the '
                      'reference rasterises the same geometry as 48 three-channel images), random-init encoder + perceptual '
                      'net; forward+backward replayed as one HIP graph (--use_graph), clipping + NaN guard + AdamW as three '
                      'multi-tensor HIP launches'}


def optimize_dp_leg(dev, iters, rank, world, dist):
    """north_star's ">= 0.9 scaling of the gradient all-reduce" is about THIS message: the ~57 MB of DDP gradients of the
    optimisation step (nnutils/train_utils.py:104-109, :277 of the reference), not the 29 KB mesh gradient of the raster
    micro-benchmark.  Every rank runs optimize.py's step on its own frame pairs (the trainer shards them like
    DistributedSampler: weak scaling), forward + backward replayed as HIP graphs, gradients averaged over the ranks -- once
    with the all-reduce of the late gradients overlapped with the rest of the backward replay (the default), once as one flat
    message after it (--nooverlap_allreduce).  Per variant: step ms (barrier on both sides, max over ranks), the collective's
    span and its exposed part from HIP events on the trainer's stream (max over ranks of the per-rank means), iterations/s."""
    import optimize
    from lasr_amd.nnutils import train_utils
    out = {'world_size': world, 'iters': iters,
           'config': 'spot3 stage 0 per rank (batch 1 pair, n_hypo 8, n_bones 21, V=642/F=1280, 256x256), --use_graph; weak scaling: '
                     'global batch = world_size pairs; random-init encoder / perceptual net, synthetic 3-frame sequence'}
    for name, extra in (('overlap', []), ('no_overlap', ['--nooverlap_allreduce'])):
        opts = optimize.parse_flags(['--name', 'bench', '--checkpoint_dir', '', '--only_mean_sym', '--nouse_gtpose',
                                     '--subdivide', '3', '--n_bones', '21', '--n_hypo', '8', '--num_epochs', '5',
                                     '--batch_size', '1', '--opt_tex', 'yes', '--iters_per_epoch', str(iters + 6), '--use_graph'] + extra)
        opts.local_rank = dev.index
        torch.manual_seed(0)
        tr = train_utils.LASRTrainer(opts).init_training()
        tr.model.train()
        tr.reinit_bones()
        for i in range(6):
            tr.module.iters = i
            tr.train_step(tr.set_input(tr.dataloader[i]))
        nbytes = sum(p.grad.numel() * 4 for p in tr.module.parameters() if p.grad is not None)
        tr.time_comm = True
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(iters):
            tr.module.iters = 6 + i
            loss, _ = tr.train_step(tr.set_input(tr.dataloader[6 + i]))
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ct = tr.comm_times_ms()
        span = sum(c[0] for c in ct) / max(len(ct), 1)
        exposed = sum(c[1] for c in ct) / max(len(ct), 1)
        t = torch.tensor([dt, span, exposed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, span, exposed = (float(x) for x in t.tolist())
        out[name] = {'ms_per_iter': dt / iters * 1e3, 'iters_per_s': iters / dt, 'pairs_per_s': world * iters / dt,
                     'allreduce_span_ms': span, 'allreduce_exposed_ms': exposed, 'grad_message_bytes': nbytes,
                     'allreduce_busbw_GBs': (2.0 * (world - 1) / world * nbytes / (span * 1e-3) / 1e9) if span > 0 else None,
                     'final_loss': float(loss)}
        del tr
        torch.cuda.synchronize()
    return out


STEP_CONFIGS = {
    # scripts/spot3.sh:24 -- stage 0: 1 pair per GPU, 8 hypotheses, 21 bones, icosphere-3, 256x256: 16 meshes per render
    'spot3_s0': dict(flags=['--only_mean_sym', '--subdivide', '3', '--n_bones', '21', '--n_hypo', '8', '--batch_size', '1'],
                     note='spot3 stage 0 (scripts/spot3.sh:24): B=1 pair, H=8, K=21, V=642/F=1280, 256x256, 16 meshes per render'),
    # scripts/template.sh:30 -- last stage of the camel schedule: 2 pairs per GPU, 1 hypothesis, 36 bones, 2560 faces, 512x512,
    # symmetry constraint dropped (the mesh comes out of the stage hand-off's exact-count re-mesh)
    'camel_s4': dict(flags=['--nosymmetric', '--noonly_mean_sym', '--subdivide', '3', '--n_bones', '36', '--n_hypo', '1', '--batch_size', '2',
                            '--img_size', '512', '--n_faces', '2560', '--n_frames', '8'],
                     stage_before=['--subdivide', '3', '--n_bones', '1', '--n_hypo', '1', '--batch_size', '1', '--img_size', '512',
                                   '--n_frames', '8'],
                     note='camel stage 4 (scripts/template.sh:30, BASELINE configs[2]): B=2 pairs, H=1, K=36, V=1282/F=2560 (re-meshed), '
                          '512x512, 4 meshes per render'),
}
RASTER_KERNELS = ('sr_forward_kernel', 'sr_backward_kernel', 'sr_setup_kernel', 'sr_tile_weight_kernel', 'sr_order_kernel',
                  'sr_forward_coop_kernel', 'sr_forward_pairs_kernel', 'sr_forward_pairs3_kernel', 'sr_forward_pairs_teams_kernel', 'sr_forward_seg_kernel')


def step_worker(name):
    """Child process of in_scope_step_leg (run under rocprofv3 --kernel-trace): a few optimisation iterations of one
    configuration, forward + backward replayed as a HIP graph like optimize.py runs them."""
    import tempfile
    import optimize
    from lasr_amd.nnutils import train_utils
    cfg = STEP_CONFIGS[name]
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    common = ['--nouse_gtpose', '--num_epochs', '5', '--opt_tex', 'yes', '--iters_per_epoch', '40', '--use_graph']
    extra = []
    with tempfile.TemporaryDirectory() as tmp:
        if 'stage_before' in cfg:                      # the previous stage's checkpoint for the hand-off (train_utils.load_network)
            o = optimize.parse_flags(['--name', 'prev', '--checkpoint_dir', tmp] + common + cfg['stage_before'])
            o.local_rank = 0
            torch.manual_seed(0)
            tr0 = train_utils.LASRTrainer(o).init_training()
            tr0.epoch_nscore = torch.zeros(o.n_hypo, device=dev)
            tr0.save('latest')
            extra = ['--model_path', os.path.join(tr0.save_dir, 'pred_net_latest.pth')]
            del tr0
        opts = optimize.parse_flags(['--name', 'bench', '--checkpoint_dir', ''] + common + cfg['flags'] + extra)
        opts.local_rank = 0
        torch.manual_seed(0)
        tr = train_utils.LASRTrainer(opts).init_training()
    tr.model.train()
    tr.reinit_bones()
    n = len(tr.dataloader)
    for i in range(18):                                # iteration 0: eager (part render); then capture; then replays
        tr.module.iters = i
        tr.train_step(tr.set_input(tr.dataloader[i % n]))
    torch.cuda.synchronize()
    print('step-worker %s done' % name, flush=True)


def _demangled_base(name):
    """'_ZN4lasr17sr_forward_kernelILb1E...' / 'void lasr::sr_forward_kernel<...>(...)' -> ('sr_forward_kernel', in_library)."""
    import re
    m = re.match(r'_ZN4lasr(\d+)', name)
    if m:
        k = int(m.group(1))
        return name[m.end():m.end() + k], True
    m = re.match(r'_Z(\d+)', name)
    if m:
        k = int(m.group(1))
        base = name[m.end():m.end() + k]
        return base, base == 'gather_rows_kernel'
    m = re.search(r'lasr::([A-Za-z0-9_]+)', name)
    if m:
        return m.group(1), True
    if 'gather_rows_kernel' in name:
        return 'gather_rows_kernel', True
    return name.split('(')[0][:60], False


def step_trace_summary(db_path, cfg_name):
    """One steady-state iteration (between two tail_adamw launches) of a rocprofv3 rocpd kernel trace -> per-kernel us of the
    library's kernels, grouped raster / tail / other, and what remains (MIOpen convolutions, torch glue of the networks)."""
    import sqlite3
    c = sqlite3.connect(db_path)
    rows = list(c.execute('select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d '
                          'join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start'))
    ends = [i for i, r in enumerate(rows) if 'tail_adamw' in r[0]]
    if len(ends) < 3:
        raise RuntimeError('the trace holds %d optimisation steps' % len(ends))
    per = {}
    n_iters = min(5, len(ends) - 1)                    # average over the last few iterations
    for j in range(n_iters):
        for name, t0, t1 in rows[ends[-2 - j] + 1:ends[-1 - j] + 1]:
            base, ours = _demangled_base(name)
            e = per.setdefault(base, [0.0, 0, ours])
            e[0] += (t1 - t0) / 1e3
            e[1] += 1
    groups = {'raster': {}, 'tail': {}, 'other_in_scope': {}}
    out_scope_us, out_scope_n = 0.0, 0
    for base, (us, cnt, ours) in per.items():
        us, cnt = us / n_iters, cnt / n_iters
        if not ours:
            out_scope_us += us
            out_scope_n += cnt
            continue
        g = 'raster' if base in RASTER_KERNELS else ('tail' if base.startswith('tail_') else 'other_in_scope')
        groups[g][base] = {'us': round(us, 2), 'launches': round(cnt, 2)}
    seq = rows[ends[-2] + 1:ends[-1] + 1]
    res = {'config': STEP_CONFIGS[cfg_name]['note'], 'iterations_averaged': n_iters,
           'kernels_per_iteration': len(seq), 'wall_us': round((seq[-1][2] - seq[0][1]) / 1e3, 1)}
    for g, ks in groups.items():
        res[g + '_us'] = round(sum(k['us'] for k in ks.values()), 1)
        res[g + '_launches'] = round(sum(k['launches'] for k in ks.values()), 1)
        res[g] = dict(sorted(ks.items(), key=lambda kv: -kv[1]['us']))
    res['out_of_scope_us'] = round(out_scope_us, 1)
    res['out_of_scope_launches'] = round(out_scope_n, 1)
    return res


def loss_reduction_figures(step, I, H, P, feat_shapes):
    """GB/s of the loss-reduction kernels north_star names, from the same trace: algorithmic bytes (silhouette 12 P,
    flow 24 P, texture 40 P bytes per rendered image = 76 P for the fused table kernels; the backward reads the same planes again
    and writes the ten gradient planes: 116 P; perceptual reduction 2 / 3 x N C P x 4 per layer, fused.hip) / kernel time."""
    N = I * H
    ks = step['other_in_scope']
    out = {}

    def add(name, kernels, nbytes):
        us = sum(ks[k]['us'] for k in kernels if k in ks)
        if us > 0:
            out[name] = {'kernels': [k for k in kernels if k in ks], 'us': round(us, 2), 'algorithmic_bytes': int(nbytes),
                         'GBs': round(nbytes / us / 1e3, 1), 'frac_of_hbm_peak': round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)}
    add('render_tables_forward', ['render_tables_forward_kernel', 'render_tables_flow_kernel', 'render_tables_fold_kernel',
                                  'render_tables_flow_fold_kernel'], 76 * P * N)
    add('render_tables_backward', ['render_tables_backward_kernel', 'render_tables_intrinsics_fold_kernel'], 116 * P * N)
    cp = sum(c * p for c, p in feat_shapes)
    add('cosdist_forward', ['cosdist_multi_forward_kernel', 'cosdist_forward_kernel', 'cosdist_fold_kernel'], 2 * 2 * N * cp * 4)
    add('cosdist_backward', ['cosdist_multi_backward_kernel', 'cosdist_backward_kernel'], 3 * 2 * N * cp * 4)
    return out


def in_scope_step_leg():
    """where the optimisation step's in-scope time goes at the sizes LASR launches.  For each configuration a
    child process runs a few graph-replayed iterations under `rocprofv3 --kernel-trace`; the trace's dispatch timestamps (what
    profiles/r06_optimize_step_kernel_stats.txt is built from) give per-kernel us of one iteration.  In-library HIP events cannot
    time kernels inside a graph replay, and around eager launches they add the launch gap to every few-us kernel."""
    import glob
    import shutil
    import subprocess
    import tempfile
    out = {'source': 'rocprofv3 --kernel-trace of `bench.py --step-worker <config>` (child process), dispatch end - start per kernel, '
                     'mean of the last iterations; groups: raster = sr_* kernels, tail = clip / NaN guard / AdamW, other_in_scope = '
                     'every other kernel of liblasr_hip.so; out_of_scope = MIOpen / rocBLAS / ATen kernels of the encoder and AlexNet'}
    if shutil.which('rocprofv3') is None:
        out['error'] = 'rocprofv3 not on PATH'
        return out
    if under_profiler():
        out['error'] = 'skipped: bench.py is itself running under rocprofv3'
        return out
    shapes = {'spot3_s0': (2, 8, 256 * 256), 'camel_s4': (4, 1, 512 * 512)}
    for name in STEP_CONFIGS:
        tmp = tempfile.mkdtemp(prefix='lasr_step_', dir='/tmp')
        try:
            env = dict(os.environ, TMPDIR='/tmp')
            cmd = ['rocprofv3', '--kernel-trace', '-d', tmp, '-o', 't', '--', sys.executable, os.path.abspath(__file__), '--step-worker', name]
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420)
            dbs = glob.glob(os.path.join(tmp, '**', '*.db'), recursive=True)
            if r.returncode != 0 or not dbs:
                out[name] = {'error': 'rc %d: %s' % (r.returncode, r.stdout.decode(errors='replace')[-400:])}
                continue
            step = step_trace_summary(dbs[0], name)
            I, H, P = shapes[name]
            side = int(P ** 0.5)
            f = lambda n, k, s_, p_: (n + 2 * p_ - k) // s_ + 1                     # noqa: E731
            c1 = f(side, 11, 4, 2)
            c2 = f(c1, 3, 2, 0)
            c3 = f(c2, 3, 2, 0)
            step['loss_reductions'] = loss_reduction_figures(step, I, H, P, [(64, c1 * c1), (192, c2 * c2), (384, c3 * c3),
                                                                             (256, c3 * c3), (256, c3 * c3)])
            out[name] = step
        except Exception as e:                       # the headline line must not depend on the profiler
            out[name] = {'error': repr(e)[:400]}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


class LineOnce:
    """Rank 0's single JSON line, printable from the main thread or from the watchdog below, whichever comes first."""

    def __init__(self):
        import threading
        self._gate, self._done = threading.Lock(), False

    def __call__(self, obj):
        with self._gate:
            if self._done:
                return False
            self._done = True
        print(json.dumps(obj), flush=True)
        return True


def run_optional_collective_legs(fn, rank, out, key, limit_s, emit):
    """Run the legs that only exist for N > 1 (collectives between the ranks: nothing this builder could execute on more than one GPU)
    so that the contract line cannot be lost to them.  If `fn` raises on this rank, or has not returned after limit_s seconds (a rank
    that failed leaves the others waiting in a collective), rank 0 prints the line measured so far with {key: {'error': ...}} and
    every rank leaves with os._exit(0) -- no communicator tear-down, which could block on the same dead collective.
    Returns fn()'s value otherwise."""
    import threading

    def give_up(reason):
        if rank == 0 and out is not None:
            out[key] = {'error': reason}
            emit(out)
        sys.stdout.flush()
        os._exit(0)
    timer = threading.Timer(limit_s, give_up, ('did not finish within %.0f s on rank %d (LASR_BENCH_DP_TIMEOUT)' % (limit_s, rank),))
    timer.daemon = True
    timer.start()
    try:
        value = fn()
    except Exception as e:                                   # noqa: BLE001 -- whatever it is, the headline line goes out
        timer.cancel()
        give_up('failed on rank %d: %s' % (rank, repr(e)[:300]))
    timer.cancel()
    return value


def allreduce_variants_leg(dev, nbytes, world, dist, reps=5):
    """how the optimisation step's gradient message (nbytes, fp32) crosses the xGMI mesh, three ways, so that
    the first 8-GPU run of this line can compare them (SURVEY section 5 predicts ring ~0.65 ms vs direct reduce-scatter /
    all-gather ~0.09 ms for 57 MB on the 7-link full mesh):
      one_all_reduce            one flat message (what the trainer sends without overlap)
      reduce_scatter_all_gather the same reduction as its two halves: every link busy in both
      buckets_25MB              DDP-sized buckets back to back (the reference's transport, nnutils/train_utils.py:104-109)
    Each: HIP events on the calling stream around `reps` repetitions after 2 warm-ups, MAX over ranks; algbw = bytes / time,
    busbw = algbw x 2 (world - 1) / world.  RCCL's own choice can be steered from outside with NCCL_ALGO / NCCL_PROTO (recorded)."""
    n = max(world, nbytes // 4 // world * world)                 # floats, a multiple of the world size
    buf = torch.ones(n, dtype=torch.float32, device=dev)
    shard = torch.empty(n // world, dtype=torch.float32, device=dev)
    bucket = max(world, (25 << 20) // 4 // world * world)

    def one():
        dist.all_reduce(buf)

    def rs_ag():
        dist.reduce_scatter_tensor(shard, buf)
        dist.all_gather_into_tensor(buf, shard)

    def buckets():
        for o in range(0, n, bucket):
            dist.all_reduce(buf[o:o + bucket])
    out = {'message_bytes': n * 4, 'world_size': world, 'reps': reps,
           'env': {k: os.environ.get(k) for k in ('NCCL_ALGO', 'NCCL_PROTO', 'NCCL_MIN_NCHANNELS', 'NCCL_MAX_NCHANNELS',
                                                  'RCCL_MSCCL_ENABLE', 'HSA_ENABLE_IPC_MODE_LEGACY')}}
    for name, fn in (('one_all_reduce', one), ('reduce_scatter_all_gather', rs_ag), ('buckets_25MB', buckets)):
        try:
            for _ in range(2):
                fn()
                buf.fill_(1.)
            dist.barrier()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) / reps], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t.item())
            out[name] = {'ms': ms, 'algbw_GBs': n * 4 / ms / 1e6, 'busbw_GBs': n * 4 / ms / 1e6 * 2 * (world - 1) / world,
                         'collectives_per_message': {'one_all_reduce': 1, 'reduce_scatter_all_gather': 2}.get(name, -(-n // bucket))}
        except Exception as e:                           # e.g. gloo has no reduce_scatter: the block keeps its shape
            out[name] = {'error': repr(e)[:200]}
        buf.fill_(1.)
    return out


RASTER_SOURCES = ('sr_raster.hip', 'sr_forward_coop.h', 'sr_forward_pairs.h', 'sr_device.h', 'sr_common.h', 'sr_backward.h', 'sr_backward_fast.hip', 'Makefile')
VALU_PEAK_LANE_OPS = 1024 * 32 * 2.4e9      # 256 CUs x 4 SIMDs, 32 fp32 lanes per SIMD per clock (a wave64 op issues in 2), 2.4 GHz


def raster_source_hash():
    """sha256 over the raster kernel sources + build flags; tools/{traffic,valu}_json.py store it in the counter files they
    write, so a committed PMC pass that predates a kernel change is recognised (and not reported) instead of going stale
    silently.  tests/test_profiles_fresh.py fails when the newest committed files do not match."""
    import hashlib
    h = hashlib.sha256()
    for n in RASTER_SOURCES:
        h.update(open(os.path.join(ROOT, 'lasr_amd', 'csrc', n), 'rb').read())
    return h.hexdigest()[:16]


def _latest_profile(suffix):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*' + suffix)))
    if not files:
        return None, None, 'no profiles/*%s committed' % suffix
    d = json.load(open(files[-1]))
    name = os.path.basename(files[-1])
    if d.get('source_sha') != raster_source_hash():
        msg = ('%s was measured on other raster sources (its source_sha %s, now %s): re-run tools/prof/r06_final.sh'
               % (name, d.get('source_sha'), raster_source_hash()))
        print('bench.py: STALE COUNTER FILE -- ' + msg, file=sys.stderr, flush=True)
        return None, name, msg
    return d, name, None


def measured_traffic(kernel, frames_per_launch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_traffic.json, latest round), quoted when
    this run launches the same number of frames.  PMC counters cannot be read from inside this process; the file records the
    exact command they came from.  None if no profile has been committed."""
    d, name, stale = _latest_profile('_traffic.json')
    if d is None:
        return None, name, stale
    k = d.get('kernels', {}).get(kernel)
    if not k:
        return None, name, 'kernel not in ' + name
    if frames_per_launch != d['frames_per_launch']:      # another launch size may run another forward kernel (forward_impl)
        return None, name, 'counters were taken at %d frames per launch' % d['frames_per_launch']
    return k['bytes'], name, None


def valu_issue(kernel, frames_per_launch, launch_ms):
    """What actually bounds the raster kernels: VALU issue.  Instruction counts of one launch from the committed SQ counter
    passes (profiles/*_valu.json, built by tools/valu_json.py from rocprofv3 --pmc SQ_INSTS_VALU* / SQ_THREAD_CYCLES_VALU),
    combined with THIS run's launch time:
      valu_frac          = live lane-operations per second / (1024 SIMDs x 32 lanes x 2.4 GHz)   -- the tracked fraction
      frac_of_launch_time = instruction mix priced with the issue costs measured by tools/ubench/ / launch time."""
    d, name, stale = _latest_profile('_valu.json')
    if d is None:
        return {'source': name, 'stale': stale}
    k = d.get('kernels', {}).get(kernel)
    if not k:
        return {'source': name, 'stale': 'kernel not in ' + name}
    if frames_per_launch != d['frames_per_launch']:      # another launch size may run another forward kernel (forward_impl)
        return {'source': name, 'stale': 'counters were taken at %d frames per launch' % d['frames_per_launch']}
    scale = 1.0
    lane_ops = k['valu_wave_instructions'] * scale * 64 * k['live_lane_fraction']
    return {'source': name, 'valu_wave_instructions_per_launch': k['valu_wave_instructions'] * scale,
            'live_lane_fraction': k['live_lane_fraction'],
            'useful_lane_ops_per_s': lane_ops / (launch_ms * 1e-3), 'peak_lane_ops_per_s': VALU_PEAK_LANE_OPS,
            'valu_frac': lane_ops / (launch_ms * 1e-3) / VALU_PEAK_LANE_OPS,
            'issue_slot_frac': k['valu_wave_instructions'] * scale * 2 / (1024 * 2.4e9 * launch_ms * 1e-3),
            'wait_any_fraction_of_wave_cycles': k.get('wait_any_fraction_of_wave_cycles'),
            'modelled_issue_ms': k['modelled_issue_ms'] * scale,
            'frac_of_launch_time': k['modelled_issue_ms'] * scale / launch_ms,
            'note': 'valu_frac counts live lanes only (an instruction with 28 of 64 lanes enabled is 28 lane-ops); '
                    'issue_slot_frac = wave instructions x 2 cycles / SIMD cycles; modelled_issue_ms prices the mix with the '
                    'measured per-instruction costs (SGPR-operand and transcendental ops issue slower); packed fp32 and MFMA '
                    'do not apply to this arithmetic (DESIGN.md section 4)'}


def variants_leg(dev, B):
    """The other points of SURVEY section 8(d) at the headline launch size: sigma = 1e-5 (the last stage of the reference's schedule,
    scripts/template.sh:32: a 1.2-pixel reject radius instead of 3.9, shorter lists) and Nf = 3 (the 3-frame cycle of syn-spot3f,
    scripts/render_syn.py:31,35: three distinct poses repeated).  Same step as `value`."""
    global IS
    keep = IS
    out = {}
    h = _lib.lib()
    for name, kw in (('sigma_1e-5', dict(sigma=1e-5)), ('nf_3', dict(n_cycle=3))):
        job = RasterStep(dev, B, 0, image_size=keep, **kw)
        for _ in range(3):
            job.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            job.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        h.lasr_prof_enable(job.stream, 1)
        for _ in range(5):
            job.step()
        torch.cuda.synchronize()
        h.lasr_prof_enable(job.stream, 0)
        out[name] = {'frames_per_s': B / dt, 'ms_per_step': dt * 1e3,
                     'kernel_ms': {k: round(v[0], 5) for k, v in collect_kernel_times(h, job.stream).items()}}
        del job
    IS = keep
    return out


def sweep_leg(dev, points, steps_budget_ms=150.0):
    """The launch sizes the reference actually uses (N = 2 / 4 / 16 / 96 meshes per render call), through the
    same RasterStep as `value`: frames/s, us per frame and per-kernel ms (library HIP events) at each (frames, image size)."""
    global IS
    keep = IS
    out = []
    h = _lib.lib()
    for B, size in points:
        job = RasterStep(dev, B, 0, image_size=size)
        for _ in range(3):
            job.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        job.step()
        torch.cuda.synchronize()
        steps = int(max(5, min(200, steps_budget_ms * 1e-3 / max(time.perf_counter() - t0, 1e-5))))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            job.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        h.lasr_prof_enable(job.stream, 1)
        for _ in range(min(steps, 10)):
            job.step()
        torch.cuda.synchronize()
        h.lasr_prof_enable(job.stream, 0)
        kt = collect_kernel_times(h, job.stream)
        seg = None
        if B <= 16:
            # opt-in LASR_SR_SEGMENTED (a tile's list folded by 4 / 8 waves in parallel, partial states merged: another rounding
            # sequence, image within 1e-6 of the default): the forward kernel alone and the step, at the launch sizes it is for
            job.forward_flags = _lib.SR_SEGMENTED
            for _ in range(3):
                job.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                job.step()
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t1) / steps
            h.lasr_prof_enable(job.stream, 1)
            for _ in range(min(steps, 10)):
                job.step()
            torch.cuda.synchronize()
            h.lasr_prof_enable(job.stream, 0)
            seg = {'frames_per_s': B / dts, 'forward_kernel_ms': round(collect_kernel_times(h, job.stream)['sr_forward_kernel'][0], 5)}
            job.forward_flags = 0
        out.append({'frames': B, 'image_size': size, 'frames_per_s': B / dt, 'ms_per_step': dt * 1e3, 'segmented_opt_in': seg,
                    'us_per_frame': dt / B * 1e6, 'steps': steps,
                    'kernel_ms': {k: round(v[0], 5) for k, v in kt.items()},
                    'kernel_us_per_frame': round(sum(v[0] * v[1] for k, v in kt.items()) / min(steps, 10) / B * 1e3, 3)})
        del job
    IS = keep
    return out


def main():
    global IS, REBUILD_RECORDS
    a = parse()
    if a.step_worker:
        return lbs_worker() if a.step_worker == 'lbs' else step_worker(a.step_worker)
    IS = a.image_size
    REBUILD_RECORDS = a.rebuild_records
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != a.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no HIP device visible); there is no CPU fallback')
    backend = os.environ.get('LASR_BENCH_BACKEND', 'nccl')       # 'nccl' == RCCL on ROCm.  'gloo': functional check of the
    if backend != 'nccl':                                        # multi-rank path with all ranks sharing GPU 0 (1-GPU box)
        local = 0
    elif a.gpus > torch.cuda.device_count():
        raise SystemExit('--gpus %d but only %d HIP device(s) visible' % (a.gpus, torch.cuda.device_count()))
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        # called as `python bench.py --gpus N`: launch one process per GPU ourselves (scripts/template.sh:26 does the same
        # with torch.distributed.launch) and relay rank 0's JSON line
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    B = a.frames
    # frames shard across ranks: rank r renders yaw positions r*B .. r*B+B-1 of the cycle (weak scaling)
    job = RasterStep(dev, B, first_frame=rank * B)

    def one_step():
        mg = job.step()
        parallel.allreduce_grads_([mg], average=False)   # mesh-parameter gradient, [2,V,3] fp32, one RCCL message
        return mg

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    device_ids = [torch.cuda.current_device()]
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ids = [None] * world
        dist.all_gather_object(ids, (rank, torch.cuda.current_device(), os.getpid()))
        device_ids = [d for _, d, _ in sorted(ids)]
        assert len({p for _, _, p in ids}) == world          # one process per rank

    # ---- roofline leg: per-kernel HIP-event timing inside the library (separate pass) ----
    h = job.h
    h.lasr_prof_enable(job.stream, 1)
    for _ in range(min(a.steps, 10)):
        job.step()
    torch.cuda.synchronize()
    h.lasr_prof_enable(job.stream, 0)
    ktimes = collect_kernel_times(h, job.stream)

    if rank == 0:
        F, P = job.F, IS * IS
        alg = {'sr_forward_kernel': (72 * F + 24 * P) * B, 'sr_backward_kernel': (144 * F + 40 * P) * B,
               'sr_setup_kernel': (36 + 192 + 8) * F * B}
        dom = max((k for k in ktimes if k in ('sr_forward_kernel', 'sr_backward_kernel')), key=lambda k: ktimes[k][0])
        achieved = alg[dom] / (ktimes[dom][0] * 1e-3) / 1e9
        traffic, traffic_src, traffic_stale = measured_traffic(dom, B) if IS == 256 else (None, None, 'PMC passes are taken at 256x256')
        vi = valu_issue(dom, B, ktimes[dom][0]) if IS == 256 else None
        frames = world * B * a.steps
        out = {
            'metric': 'rasterizer fwd+bwd frames/sec at %dx%d, 2.3k faces' % (IS, IS),
            'value': frames / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'world_size': world, 'backend': ('rccl (torch.distributed "nccl")' if backend == 'nccl' else backend) if world > 1 else None,
            'rank_device_ids': device_ids,
            'config': {'workload': 'soft-rasteriser fwd+bwd, mesh M2 (V=1212,F=2420), %dx%d, LASR modes '
                                   '(euclidean/softmax/prod/vertex, sigma=1e-4, gamma=1e-2)' % (IS, IS),
                       'frames_per_gpu_per_step': B, 'image_size': IS, 'faces': F, 'vertices': job.V,
                       'parallelism': 'dp%d (frames sharded, mesh-gradient all-reduce)' % world,
                       'step_definition': 'face setup + [5 frames and more: the launch\'s own tile order, sr_tile_weight_kernel + sr_order_kernel] + '
                                          'forward kernel (background colour passed as an argument: no pre-fill pass, '
                                          'every element of soft_colors written) + backward kernel on the forward\'s face records, as the autograd operator '
                                          'runs it (--rebuild-records 1: a second face setup first) (stores every '
                                          'gradient element: no zero-fill pass) + face->vertex reduction of both gradients with the product\'s deterministic '
                                          'lasr_face_gather_backward_csr (the vertex -> corner incidence of the fixed connectivity is built once, as '
                                          'the operator caches it for a face tensor it sees again) + sum over the frames '
                                          '(+ RCCL all-reduce of the [2,V,3] mesh gradient for N > 1); rounds 1 and early 2 '
                                          'also timed the two fill passes the reference caller needs (soft_rasterize.py:50-53, '
                                          ':88-89), which these entry points make unnecessary'},
            # achieved / peak / frac are the HBM figures SURVEY 8(d) defines (algorithmic bytes / kernel time); `bound` is what
            # the counters say binds the kernel: VALU issue (valu_frac is the fraction to track), HBM is two orders of magnitude away
            'roofline': {'bound': 'valu' if vi and vi.get('valu_frac') else 'hbm', 'kernel': dom, 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'traffic_stale': traffic_stale,
                         'valu_frac': vi.get('valu_frac') if vi else None,
                         'algorithmic_bytes_per_launch': alg[dom], 'avg_launch_ms': ktimes[dom][0],
                         'all_kernels_avg_ms': {k: v[0] for k, v in ktimes.items()},
                         'raster_source_sha': raster_source_hash(),
                         'valu_issue': vi},
        }
        if world == 1:
            # informational: the opt-in relaxed forward arithmetic (per-call flag LASR_SR_RELAXED_MATH, image within ~1e-5 of
            # the default path; `value` above is the default, reference-faithful arithmetic)
            job.forward_flags = _lib.SR_RELAXED_MATH
            try:
                job.step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(min(a.steps, 10)):
                    job.step()
                torch.cuda.synchronize()
                out['relaxed_forward_math'] = {'value': B * min(a.steps, 10) / (time.perf_counter() - t1), 'unit': 'frames/s',
                                               'note': 'opt-in; distance + threshold decision bit-faithful, the rest fp32 rcp/exp'}
            finally:
                job.forward_flags = 0
        def optional(key, fn):
            # the informational legs after the timed region: one that fails leaves {'error': ...} in its block, the line still goes out
            try:
                out[key] = fn()
            except Exception as e:                           # noqa: BLE001
                out[key] = {'error': repr(e)[:400]}
        if world == 1 and not a.no_sweep:
            optional('sweep', lambda: sweep_leg(dev, [(1, 256), (4, 256), (16, 256), (64, 256), (64, 512)]))
            IS = a.image_size
            optional('other_points', lambda: variants_leg(dev, B))
        if world == 1 and not a.no_cpu_baseline:
            optional('cpu_baseline', lambda: cpu_baseline(F))
        if world == 1 and not a.no_lbs:
            optional('lbs', lambda: lbs_leg(dev))
        if world == 1 and a.lasr_iters > 0:
            optional('optimize_py', lambda: optimize_leg(dev, a.lasr_iters))
        if world == 1 and a.lasr_iters > 0 and not a.no_step_profile:
            # after every timed leg: the child process shares this GPU while it runs
            torch.cuda.synchronize()
            out['in_scope_step'] = in_scope_step_leg()
    emit = LineOnce()
    dp = None
    if world > 1 and a.lasr_iters > 0:                       # every rank takes part

        def dp_legs():
            d = optimize_dp_leg(dev, a.lasr_iters, rank, world, dist)
            d['allreduce_variants'] = allreduce_variants_leg(dev, d['no_overlap']['grad_message_bytes'], world, dist)
            return d
        dp = run_optional_collective_legs(dp_legs, rank, out if rank == 0 else None, 'optimize_py_dp',
                                          float(os.environ.get('LASR_BENCH_DP_TIMEOUT', '600')), emit)
    if rank == 0:
        if dp is not None:
            out['optimize_py_dp'] = dp
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
