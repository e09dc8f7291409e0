"""bench.py's guard around the legs that only run for N > 1 (collectives the builder never executed on more than one GPU): the
contract line must survive a leg that raises or hangs.  Runs the helper in child processes on the CPU."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, time, json
sys.path.insert(0, %r)
import bench
mode, rank = sys.argv[1], int(sys.argv[2])
out = {'metric': 'm', 'value': 1.0} if rank == 0 else None
emit = bench.LineOnce()
def leg():
    if mode == 'hang':
        time.sleep(60)
    if mode == 'raise':
        raise RuntimeError('rccl said no')
    return {'ok': 1}
dp = bench.run_optional_collective_legs(leg, rank, out, 'optimize_py_dp', 1.5, emit)
if rank == 0:
    out['optimize_py_dp'] = dp
    emit(out)
    emit(out)                      # a second call prints nothing
print('after', flush=True)
'''


def run(mode, rank):
    t0 = time.time()
    p = subprocess.run([sys.executable, '-c', CHILD % ROOT, mode, str(rank)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=120, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    return p.returncode, lines, 'after' in p.stdout, time.time() - t0, p.stderr[-500:]


def test_a_normal_leg_returns_its_value_and_the_line_is_printed_once():
    rc, lines, after, _, err = run('ok', 0)
    assert rc == 0 and after and len(lines) == 1, err
    assert json.loads(lines[0]) == {'metric': 'm', 'value': 1.0, 'optimize_py_dp': {'ok': 1}}
    rc, lines, after, _, _ = run('ok', 1)
    assert rc == 0 and after and lines == []


def test_a_leg_that_raises_costs_only_its_block():
    rc, lines, after, _, err = run('raise', 0)
    assert rc == 0 and not after and len(lines) == 1, err          # hard exit right after the line: no communicator tear-down
    d = json.loads(lines[0])
    assert d['value'] == 1.0 and 'rccl said no' in d['optimize_py_dp']['error'] and 'rank 0' in d['optimize_py_dp']['error']
    rc, lines, after, _, _ = run('raise', 3)
    assert rc == 0 and not after and lines == []                    # other ranks leave quietly


def test_a_leg_that_hangs_is_cut_off_by_the_watchdog():
    rc, lines, after, dt, err = run('hang', 0)
    assert rc == 0 and not after and len(lines) == 1 and dt < 40, (dt, err)
    assert 'did not finish within' in json.loads(lines[0])['optimize_py_dp']['error']
    rc, lines, after, dt, _ = run('hang', 5)
    assert rc == 0 and not after and lines == [] and dt < 40
