#!/usr/bin/env python3
"""Where the gradient all-reduce sits in a graph-replay data-parallel step (two ranks on the one GPU of the test box, gloo; on a
node: one rank per GPU over RCCL).  Rank 0 brackets the two graph replays with HIP events and the all-reduce calls with host
timestamps (aligned to the event clock at the start of the step) and prints one line per span.

    python tools/dp_overlap_timeline.py > profiles/rNN_dp_overlap_timeline.txt
"""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import optimize
    from lasr_amd import parallel
    from lasr_amd.nnutils import train_utils
    opts = optimize.parse_flags(['--name', 'dp', '--checkpoint_dir', '', '--only_mean_sym', '--nouse_gtpose', '--subdivide', '3',
                                 '--n_bones', '21', '--n_hypo', '8', '--batch_size', '1', '--num_epochs', '1', '--opt_tex', 'yes',
                                 '--n_frames', '4', '--iters_per_epoch', '12', '--use_graph'])
    tr = train_utils.LASRTrainer(opts).init_training()
    tr.model.train()
    tr.reinit_bones()
    for i in range(6):
        tr.module.iters = i
        tr.train_step(tr.set_input(tr.dataloader[i]))
    torch.cuda.synchronize()
    spans = []
    ref_ev, ref_t = torch.cuda.Event(enable_timing=True), [0.0]
    replay = torch.cuda.CUDAGraph.replay
    n_replay = [0]

    def timed_replay(self):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        replay(self)
        b.record()
        spans.append(('graph %s replay (GPU)' % 'AB'[min(n_replay[0], 1)], a, b))
        n_replay[0] += 1
    async_ = parallel.allreduce_grads_async
    sync_ = parallel.allreduce_grads_

    def timed_async(tensors, **kw):
        t0 = time.perf_counter()
        fin = async_(tensors, **kw)
        nbytes = sum(t.numel() for t in tensors) * 4

        def finish():
            out = fin()
            spans.append(('all-reduce #1, %.1f MB: issued .. waited for (host)' % (nbytes / 1e6), t0, time.perf_counter()))
            return out
        return finish

    def timed_sync(tensors, **kw):
        t0 = time.perf_counter()
        out = sync_(tensors, **kw)
        spans.append(('all-reduce #2, %.1f MB (host)' % (sum(t.numel() for t in tensors if t is not None) * 4 / 1e6), t0, time.perf_counter()))
        return out
    torch.cuda.CUDAGraph.replay = timed_replay
    parallel.allreduce_grads_async, parallel.allreduce_grads_ = timed_async, timed_sync
    torch.cuda.synchronize()
    dist.barrier()
    ref_ev.record()
    torch.cuda.synchronize()
    ref_t[0] = time.perf_counter()
    tr.module.iters = 6
    tr.train_step(tr.set_input(tr.dataloader[6]))
    torch.cuda.synchronize()
    if rank == 0:
        rows = []
        for name, a, b in spans:
            if isinstance(a, float):
                rows.append((name, (a - ref_t[0]) * 1e3, (b - ref_t[0]) * 1e3))
            else:
                rows.append((name, ref_ev.elapsed_time(a), ref_ev.elapsed_time(b)))
        q.put(rows)
    dist.barrier()
    dist.destroy_process_group()


def main():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    rows = q.get(timeout=300)
    for p in procs:
        p.join(60)
    print('# one graph-replay DP step of rank 0 (spot3 stage-0 configuration, 2 ranks on ONE MI355X over gloo: the collective goes')
    print('# through host memory here, so its duration says nothing about RCCL over xGMI -- the ORDER is what this shows: the first')
    print('# all-reduce (everything above the cut: mesh, bones, heads, encoder layer 4) is issued before graph B and waited for')
    print('# after it, i.e. graph B (the rest of the encoder backward) runs inside the collective.  ms from the start of the step.')
    print('# Host spans start when the HOST issued the call (it runs ahead of the GPU); a collective begins on the device once the')
    print('# stream reaches that point, i.e. all-reduce #1 at the END of graph A, all-reduce #2 at the end of graph B.')
    for name, a, b in sorted(rows, key=lambda r: r[1]):
        print('%9.3f .. %9.3f   %s' % (a, b, name))


if __name__ == '__main__':
    main()
