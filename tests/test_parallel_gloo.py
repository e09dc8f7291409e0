"""world_size-2 gloo run of the data-parallel raster path on CPU: frames shard across ranks, every rank computes the
face->vertex gradient of ITS frames (the CPU oracle stands in for the HIP kernels, which need a GPU), the mesh
gradient is all-reduced as one message, and the result must equal the single-process sum over all frames."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IS, NU, NFR = 24, 2, 4


def mesh_grad_for_frames(frames):
    from lasr_amd import synth
    from oracle import sr_oracle
    v, f, tex = synth.blobby_mesh(NU)
    V = v.shape[0]
    out = np.zeros((2, V, 3), np.float64)
    for fr in frames:
        pv = synth.frame_vertices(v, NFR, first=fr, count=1)
        near, far = synth.near_far(synth.frame_vertices(v, NFR)[:, :, 2])
        fv, ft = pv[:, f], np.broadcast_to(tex[f][None], (1,) + tex[f].shape).copy()
        kw = dict(synth.LASR_MODES, near=near, far=far)
        ref = sr_oracle.forward(fv, ft, IS, **kw)
        g = synth.upstream_grad(1, IS, seed=100 + fr)
        gf, gt = sr_oracle.backward(ref, g, IS, **kw)
        np.add.at(out[0], f.reshape(-1), gf[0].reshape(-1, 3))
        np.add.at(out[1], f.reshape(-1), gt[0].reshape(-1, 3))
    return out


def worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OMP_NUM_THREADS='1')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lasr_amd import parallel
    mine = parallel.shard(range(NFR), rank, world)
    g = torch.from_numpy(mesh_grad_for_frames(mine))
    pos, col = g[0].clone(), g[1].clone()
    parallel.allreduce_grads_([pos, col], average=False)
    q.put((rank, mine, pos.numpy(), col.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_frames_allreduce_equals_single_process():
    sys.path.insert(0, ROOT)
    from lasr_amd import parallel
    assert parallel.shard(range(5), 0, 2) == [0, 2, 4] and parallel.shard(range(5), 1, 2) == [1, 3, 0]
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    frames = sorted(sum((g[1] for g in got), []))
    assert frames == list(range(NFR))
    ref = mesh_grad_for_frames(range(NFR))
    for _, _, pos, col in got:                     # both ranks hold the same, complete gradient
        np.testing.assert_allclose(pos, ref[0], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(col, ref[1], rtol=1e-9, atol=1e-12)


def test_allreduce_is_a_noop_without_a_process_group():
    from lasr_amd import parallel
    t = torch.ones(3)
    assert parallel.allreduce_grads_([t, None])[0] is t and float(t.sum()) == 3
