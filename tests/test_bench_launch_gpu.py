"""bench.py as the driver calls it: `python bench.py --gpus N` with no torchrun around it must launch one process per
rank itself (the reference does so from its script line, scripts/template.sh:26 / optimize.py:42-47) and rank 0 must
print ONE well-formed JSON line.  On a 1-GPU box the two ranks share GPU 0 and talk over gloo (LASR_BENCH_BACKEND);
with >= 2 GPUs the same test runs the RCCL path."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(args, env_extra=None, timeout=600):
    env = dict(os.environ, **(env_extra or {}))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, cwd=ROOT, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_self_launch(cuda):
    backend = 'nccl' if torch.cuda.device_count() >= 2 else 'gloo'
    out = run_bench(['--gpus', '2', '--steps', '3', '--warmup', '1', '--frames', '16', '--lasr-iters', '4'],
                    {'LASR_BENCH_BACKEND': backend}, timeout=900)
    assert out['n_gpus'] == 2 and out['world_size'] == 2 and out['steps'] == 3 and out['scaling'] == 'weak'
    assert out['metric'].startswith('rasterizer fwd+bwd frames/sec at 256x256')
    assert out['value'] > 0 and out['roofline']['frac'] > 0 and len(out['rank_device_ids']) == 2
    assert ('rccl' in out['backend']) == (backend == 'nccl')
    assert abs(out['value'] - 2 * 16 * 3 / (out['ms_per_step'] * 3e-3)) <= 1e-6 * out['value']
    # the optimisation step's gradient all-reduce (the message north_star's 0.9-scaling target is about), with and without overlap
    dp = out['optimize_py_dp']
    assert dp['world_size'] == 2
    for name in ('overlap', 'no_overlap'):
        v = dp[name]
        assert v['iters_per_s'] > 0 and v['grad_message_bytes'] > 40e6 and v['allreduce_span_ms'] > 0
        assert 0 <= v['allreduce_exposed_ms'] <= v['allreduce_span_ms'] * 1.001 and np.isfinite(v['final_loss'])
        assert abs(v['pairs_per_s'] - 2 * v['iters_per_s']) <= 1e-9 * v['pairs_per_s']
    assert dp['no_overlap']['allreduce_exposed_ms'] == dp['no_overlap']['allreduce_span_ms']
    # the three transports of the gradient message, timed for the first multi-GPU run to compare (values only mean something on xGMI)
    av = dp['allreduce_variants']
    assert av['message_bytes'] > 40e6 and av['world_size'] == 2 and 'NCCL_ALGO' in av['env']
    for name in ('one_all_reduce', 'reduce_scatter_all_gather', 'buckets_25MB'):
        assert ('ms' in av[name] and av[name]['ms'] > 0 and av[name]['busbw_GBs'] > 0) or 'error' in av[name], av[name]
    assert 'ms' in av['one_all_reduce'] and av['buckets_25MB'].get('collectives_per_message', 3) >= 2


def test_bench_eight_ranks_self_launch_on_one_gpu(cuda):
    # The driver's 8-GPU line is `python bench.py --gpus 8`; no 8-GPU node is available to the builder, so the launcher
    # plumbing (self-launch, rendezvous, rank -> frames sharding, max-over-ranks timing, object gather, the data-parallel
    # optimisation leg, rank 0's single JSON line) runs here with eight ranks sharing GPU 0 over gloo, at tiny sizes.
    out = run_bench(['--gpus', '8', '--steps', '2', '--warmup', '1', '--frames', '4', '--lasr-iters', '2'],
                    {'LASR_BENCH_BACKEND': 'gloo'}, timeout=1500)
    assert out['n_gpus'] == 8 and out['world_size'] == 8 and len(out['rank_device_ids']) == 8 and out['backend'] == 'gloo'
    assert abs(out['value'] - 8 * 4 * 2 / (out['ms_per_step'] * 2e-3)) <= 1e-6 * out['value']
    assert out['config']['parallelism'].startswith('dp8')
    dp = out['optimize_py_dp']
    assert dp['world_size'] == 8 and dp['overlap']['pairs_per_s'] > 0 and dp['no_overlap']['pairs_per_s'] > 0
    av = dp['allreduce_variants']
    assert av['world_size'] == 8 and set(av) >= {'one_all_reduce', 'reduce_scatter_all_gather', 'buckets_25MB', 'env', 'message_bytes'}
    assert 'ms' in av['one_all_reduce']


def test_bench_single_gpu_line_has_every_block(cuda):
    out = run_bench(['--steps', '3', '--warmup', '1', '--frames', '32', '--lasr-iters', '0'])
    assert out['n_gpus'] == 1 and out['dtype'] == 'f32' and out['vs_baseline'] is None
    r, c, l = out['roofline'], out['cpu_baseline'], out['lbs']
    # (32 frames per launch: the committed counter files are for 256, so no VALU fraction / traffic is quoted from them)
    assert r['bound'] == 'hbm' and r['valu_frac'] is None and r['traffic'] is None and 'frames per launch' in r['traffic_stale']
    assert r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and c['one_thread_frames_per_s'] > 0
    assert l['mfma_instruction'] == 'v_mfma_f32_16x16x4_f32' and set(l['sizes']) == {'S0', 'dog15', 'batch256'}
    assert all(v['us_per_call'] > 0 and v['mfma_instructions'] > 0 for v in l['sizes'].values())


def test_rccl_executes_on_this_box(cuda):
    # One rank, backend 'nccl' (= RCCL on ROCm): communicator creation, an all-reduce of the mesh-gradient message, a barrier and
    # the object gather bench.py uses -- the calls of the N > 1 path, on the one GPU this box has (RCCL refuses two ranks on one
    # device, so the two-rank test above runs over gloo here and over RCCL wherever there are two GPUs).
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from lasr_amd import parallel
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(%d), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
assert dist.get_backend() == 'nccl'
g = torch.arange(2 * 1212 * 3, dtype=torch.float32, device=dev).reshape(2, 1212, 3)
want = g.clone()
dist.all_reduce(g)                      # goes through RCCL even with one rank
parallel.allreduce_grads_([g], average=False)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(g, want)
ids = [None]
dist.all_gather_object(ids, (0, torch.cuda.current_device(), os.getpid()))
assert ids[0][2] == os.getpid()
dist.destroy_process_group()
print('rccl ok', torch.cuda.nccl.version())
'''
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    p = subprocess.run([sys.executable, '-c', code % (ROOT, port)], cwd=ROOT, timeout=300, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0 and 'rccl ok' in p.stdout, (p.stdout[-500:], p.stderr[-3000:])


def test_bench_line_reports_the_in_scope_kernel_time_of_the_optimisation_step(cuda):
    # the driver line carries the step's in-scope kernel time from a kernel trace taken in a child process
    import shutil
    if shutil.which('rocprofv3') is None:
        pytest.skip('no rocprofv3 on this box')
    args = ['--steps', '2', '--warmup', '1', '--frames', '16', '--lasr-iters', '2', '--no-cpu-baseline', '--no-sweep']
    out = run_bench(args, timeout=900)
    s = out['in_scope_step']
    for cfg in ('spot3_s0', 'camel_s4'):
        x = s[cfg]
        assert 'error' not in x, x
        assert 20 <= x['other_in_scope_launches'] <= 80 and 50 < x['other_in_scope_us'] < 2000
        assert x['raster_us'] > 50 and x['tail_us'] > 10 and x['out_of_scope_us'] > x['other_in_scope_us']
        assert abs(sum(v['us'] for v in x['other_in_scope'].values()) - x['other_in_scope_us']) < 1.0
        lr = x['loss_reductions']
        assert set(lr) == {'render_tables_forward', 'render_tables_backward', 'cosdist_forward', 'cosdist_backward'}
        assert all(0.02 < g['frac_of_hbm_peak'] < 1.0 for g in lr.values())
    tr = out['lbs']['sizes']['S0']['trace_us']
    assert 2 < tr['forward'] < 20 and 2 < tr['backward'] < 30
    # inside a profiled process the legs that would start a profiler of their own are skipped, the line still comes out
    out = run_bench(args, {'LASR_BENCH_UNDER_PROFILER': '1'}, timeout=600)
    assert 'skipped' in out['in_scope_step']['error'] and 'skipped' in out['lbs']['trace_error']
    assert out['value'] > 0 and out['optimize_py']['iters_per_s'] > 0
