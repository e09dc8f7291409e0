"""Trainer for the per-video optimisation (reference API: /root/reference/nnutils/train_utils.py:87-487
`LASRTrainer`, base class third_party/ext_nnutils/train_utils.py:63-136): define_model / define_criterion_ddp /
init_training / train / save_network / load_network, one process per GPU, DDP over RCCL (backend 'nccl').

Differences, all behind the same calls:
  * data comes from an in-memory synthetic sequence already resident on the device (no dataset here);
  * the NaN-gradient guard of :282-291 costs one host sync per step instead of one per parameter tensor;
  * k-means bone initialisation (:243-251, kmeans_pytorch) is a small Lloyd iteration with farthest-point seeding.
"""
import os
import sys
from types import SimpleNamespace

import torch
import torch.distributed as dist
import torch.nn as nn

from . import loss_utils, mesh_net
from .. import _lib, synth_data


# --dataname values that select the in-memory synthetic sequence (lasr_amd/synth_data.py) when no configs/<name>.config exists;
# 'fashion' is the reference's flag default (nnutils/mesh_net.py:69), which names no dataset shipped with it either
SYNTHETIC_DATANAMES = {'synthetic', 'fashion'}


def kmeans(x, k, iters=20):
    """Deterministic Lloyd k-means (farthest-point seeding).  x [n,3] -> (assignment [n], centres [k,3])."""
    c = [x[0]]
    d = (x - c[0]).pow(2).sum(1)
    for _ in range(k - 1):
        c.append(x[d.argmax()])
        d = torch.minimum(d, (x - c[-1]).pow(2).sum(1))
    c = torch.stack(c)
    for _ in range(iters):
        a = (x[:, None] - c[None]).pow(2).sum(-1).argmin(1)
        for j in range(k):
            if (a == j).any():
                c[j] = x[a == j].mean(0)
    return a, c


def set_deterministic():
    """--deterministic: the same run twice gives the same numbers.  Every lasr_amd kernel is deterministic by construction
    (ordered reductions, no float atomics on the training path); what varies between runs is outside them: MIOpen's timing-based
    algorithm search (benchmark mode) and non-deterministic library kernels.  Pins: immediate-mode algorithm choice,
    deterministic convolution kernels, torch's deterministic algorithms (warn where none exists)."""
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
    torch.use_deterministic_algorithms(True, warn_only=True)


class LASRTrainer:
    def __init__(self, opts):
        self.opts = opts
        self.save_dir = os.path.join(opts.checkpoint_dir, opts.name)
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.distributed else 0
        self.world = dist.get_world_size() if self.distributed else 1
        self.device = torch.device('cuda', opts.local_rank) if torch.cuda.is_available() else torch.device('cpu')
        if self.rank == 0 and opts.checkpoint_dir:
            os.makedirs(self.save_dir, exist_ok=True)
            with open(os.path.join(self.save_dir, 'opts.log'), 'w') as f:
                for k in sorted(vars(opts)):
                    f.write('%s: %s\n' % (k, getattr(opts, k)))

    # ---- model ------------------------------------------------------------------------------
    def define_model(self):
        opts = self.opts
        self.model = mesh_net.LASR((opts.img_size, opts.img_size), opts, nz_feat=opts.nz_feat)
        if getattr(opts, 'encoder_weights', ''):
            n = mesh_net.load_resnet18_weights(self.model.encoder.resnet_conv, opts.encoder_weights)
            if self.rank == 0:
                print('[lasr_amd] encoder: %d ResNet-18 tensors from %s' % (n, opts.encoder_weights), file=sys.stderr)
        elif self.rank == 0 and opts.model_path == '':
            print('[lasr_amd] encoder: ResNet-18 trunk is RANDOMLY initialised (the reference starts from ImageNet weights; '
                  'pass --encoder_weights <local resnet18 state_dict>)', file=sys.stderr)
        if opts.model_path != '':
            self.load_network(self.model, model_path=opts.model_path)
        self.model = nn.SyncBatchNorm.convert_sync_batchnorm(self.model).to(self.device)
        # Data parallel, two ways.  Default: DistributedDataParallel as the reference (bucketed all-reduce overlapped with the
        # eager backward pass).  With --use_graph the forward + backward of a rank is ONE HIP-graph replay (DDP's reducer
        # hooks cannot live inside a capture), so the gradients are all-reduced right after the replay as one flat RCCL
        # message instead (lasr_amd/parallel.py): a 9 ms replay + one 57 MB all-reduce beats a 25 ms eager step with overlap.
        self.manual_dp = self.distributed and getattr(opts, 'use_graph', False) and self.device.type == 'cuda'
        if self.manual_dp:
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                dist.broadcast(t.data, 0)                 # what DDP's constructor does: start from rank 0's values
        elif self.distributed:
            kw = dict(device_ids=[opts.local_rank], output_device=opts.local_rank) if self.device.type == 'cuda' else {}
            self.model = nn.parallel.DistributedDataParallel(self.model, find_unused_parameters=True, **kw)
        self.define_criterion_ddp()

    @property
    def module(self):
        return self.model.module if hasattr(self.model, 'module') else self.model

    def define_criterion_ddp(self):
        m = self.module
        mean_v, _, faces = m.get_mean_shape(1)
        m.triangle_loss_fn_sr = loss_utils.LaplacianLoss(mean_v[0].cpu(), faces[0].cpu()).to(self.device)
        m.arap_loss_fn = loss_utils.ARAPLoss(mean_v[0].cpu(), faces[0].cpu()).to(self.device)
        m.flatten_loss = loss_utils.FlattenLoss(faces[0].cpu()).to(self.device)
        m.ptex_loss = mesh_net.PerceptualDistance().to(self.device) if self.opts.perceptual else None
        if m.ptex_loss is not None:
            if getattr(self.opts, 'alexnet_weights', ''):
                m.ptex_loss.load_weights(self.opts.alexnet_weights)
                m.ptex_loss.to(self.device)
            elif self.rank == 0:
                print('[lasr_amd] perceptual term: AlexNet features are RANDOM (frozen); the reference uses ImageNet weights -- '
                      'pass --alexnet_weights <local alexnet state_dict> or --noperceptual', file=sys.stderr)

    # ---- data -------------------------------------------------------------------------------
    def init_dataset(self):
        """A sequence on disk when configs/<dataname>.config names one (dataloader/vid.py:97-134), else the in-memory
        synthetic sequence (there is no dataset on the benchmark machine)."""
        opts = self.opts
        self.sequence = None
        from ..dataloader import resident, vid
        root = getattr(opts, 'data_root', '.')
        cfg_path = vid.config_path(opts, root)
        if os.path.exists(cfg_path):
            # a sequence that HAS a config must load completely: a missing frame / mask / flow file is an error, not a
            # reason to train on something else
            loader, self.n_frames_on_disk = vid.data_loader(opts, root=root)
            if self.device.type == 'cuda':         # prepare every distinct pair once and keep it in HBM
                one = LASRTrainer.__new__(LASRTrainer)
                one.opts, one.device = SimpleNamespace(batch_size=1), torch.device('cpu')
                loader = resident.ResidentLoader(loader, one._set_input_from_loader, self.device)
            self.dataloader = loader
            if self.rank == 0:
                print('[lasr_amd] data: sequence "%s" from %s (%d frames)' % (opts.dataname, cfg_path, self.n_frames_on_disk),
                      file=sys.stderr)
            return
        if opts.dataname not in SYNTHETIC_DATANAMES:
            raise FileNotFoundError('no %s for --dataname %s; the in-memory synthetic sequence is only used for --dataname %s'
                                    % (cfg_path, opts.dataname, ' / '.join(sorted(SYNTHETIC_DATANAMES))))
        if self.rank == 0:
            print('[lasr_amd] data: in-memory SYNTHETIC sequence (--dataname %s, %d frames): no dataset is read'
                  % (opts.dataname, opts.n_frames), file=sys.stderr)
        self.sequence = synth_data.SyntheticSequence(self.device, opts.img_size, n_frames=opts.n_frames)
        npairs = len(self.sequence.pairs())
        # an epoch is padded to ~200 iterations per rank (dataloader/vid.py:78-80); pairs are dealt round-robin
        # over ranks like DistributedSampler does (dataloader/vid.py:126-131)
        per_epoch = opts.iters_per_epoch
        order = torch.randperm(max(npairs, per_epoch * opts.batch_size * self.world),
                               generator=torch.Generator().manual_seed(0)) % npairs
        mine = order[self.rank::self.world].to(self.device)          # ids live on the device: a batch is a view + one gather launch
        self.dataloader = [mine[i * opts.batch_size:(i + 1) * opts.batch_size] for i in range(per_epoch)]

    def set_input(self, batch):
        if not isinstance(batch, dict):
            return self.sequence.batch(batch)                    # list of pair ids of the synthetic sequence
        if 'input_imgs  ' in batch:
            return batch                                         # ResidentLoader: already the model's dictionary, on the device
        return self._set_input_from_loader(batch)

    def _set_input_from_loader(self, batch):
        """Collated loader elements -> the model's batch dictionary (train_utils.py:125-181): frame t block then frame
        t' block, ImageNet normalisation for the encoder input, final interleave."""
        B, dev = self.opts.batch_size, self.device

        def pair(a, b):
            return torch.cat([batch[a].float(), batch[b].float()], 0).to(dev, non_blocking=True)

        def both(name, *tail):                                   # [B,2,...] -> [2B,...], frame-major
            t = batch[name].float().to(dev, non_blocking=True)
            return t.transpose(0, 1).reshape(2 * B, *tail)
        imgs = pair('img', 'imgn')
        IS = imgs.shape[-1]
        mean = imgs.new_tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
        std = imgs.new_tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        out = {
            'input_imgs  ': (imgs - mean) / std,
            'imgs        ': imgs,
            'masks       ': both('mask', IS, IS),
            'cams        ': pair('cam', 'camn'),
            'depth_gt    ': pair('depth', 'depthn').reshape(2 * B, -1),
            'flow        ': pair('flow', 'flown'),
            'dts_barrier ': both('mask_dts', 1, IS, IS),
            'ddts_barrier': both('dmask_dts', 1, IS, IS),
            'mask_contour': both('mask_contour', 1, 1000, 2),
            'pp          ': both('pps', 2),
            'occ         ': pair('occ', 'occn'),
            'oriimg_shape': batch['shape'][:1, :2].float().repeat(2 * B, 1).to(dev),
            'is_canonical': pair('is_canonical', 'is_canonicaln'),
            'frameid': pair('id0', 'id1'),
            'dataid': torch.cat([batch['dataid'][:B], batch['dataid'][:B]], 0).float().to(dev),
        }
        return {k: v.view(2, B, -1).permute(1, 0, 2).reshape(v.shape) for k, v in out.items()}

    # ---- optimisation -----------------------------------------------------------------------
    def init_training(self):
        opts = self.opts
        if getattr(opts, 'deterministic', False):
            set_deterministic()
        if getattr(opts, 'use_graph', False) and self.device.type == 'cuda':
            self._stream = torch.cuda.Stream(self.device)     # see _graphed_forward_backward
            torch.cuda.set_stream(self._stream)
        self.init_dataset()
        self.define_model()
        m = self.module
        special = ('mean_v', 'tex', 'ctl_rs', 'rest_ts', 'ctl_ts', 'log_ctl')
        rest = [p for n, p in m.named_parameters() if n not in special]
        groups = [{'params': rest}]
        for n in special:                                   # 50x learning rate for the mesh / bone parameters (:205-214)
            p = getattr(m, n)
            if isinstance(p, nn.Parameter):
                groups.append({'params': [p], 'lr': 50 * opts.learning_rate})
        # fused: one multi-tensor launch per parameter group instead of ~9 foreach launches
        self.optimizer = torch.optim.AdamW(groups, lr=opts.learning_rate, betas=(0.9, 0.999), weight_decay=1e-4,
                                           fused=self.device.type == 'cuda')
        max_lr = [opts.learning_rate] + [50 * opts.learning_rate] * (len(groups) - 1)
        self.scheduler = torch.optim.lr_scheduler.OneCycleLR(
            self.optimizer, max_lr, 200 * len(self.dataloader), pct_start=0.01, cycle_momentum=False,
            anneal_strategy='linear', final_div_factor=1. / 25)
        return self

    # ---- HIP-graph replay of forward + backward -------------------------------------------------
    # The step issues ~2.6k small kernels (ResNet-18, loss glue) and is bound by host launch cost, not by the GPU.
    # With --use_graph the forward + backward of one iteration is captured once per (epoch, configuration) into a
    # HIP graph and replayed with the batch copied into static buffers; clipping, the NaN guard, AdamW and the LR
    # schedule stay eager.  Under torch.distributed the gradients are all-reduced after the replay (define_model).
    def _graph_key(self):
        m, o = self.module, self.opts
        noisy = o.noise and m.epoch > 0 and 1 < m.iters < 100
        if not getattr(o, 'use_graph', False) or self.device.type != 'cuda':
            return None
        if m.iters == 0:                                # first iteration of an epoch: part rendering, logging only
            return None
        # epoch- and iteration-dependent scalars live in device buffers (LASR.schedule_scalars), so the op sequence only
        # depends on whether the pose-noise branch runs and on which hypothesis is rendered for the parts
        return ('noisy' if noisy else 'plain', int(m.optim_idx))

    def _graphed_forward_backward(self, batch, key):
        g = self._graphs.get(key) if hasattr(self, '_graphs') else None
        if g is None:
            persistent = getattr(batch, 'persistent', False)     # views into a buffer that never moves (dataloader/packed.py):
            if not hasattr(self, '_graphs'):                     # the graph reads its inputs straight from it
                self._graphs, self._static = {}, (batch if persistent else {k: v.clone() for k, v in batch.items()})
            if self._static is not batch:
                for k, v in batch.items():
                    self._static[k].copy_(v)
            # Everything (eager iterations included) runs on self._stream, a non-default stream chosen in
            # init_training: the captured backward then never has to hand gradients to an AccumulateGrad node that
            # lives on another stream (a cross-stream dependency inside a capture crashes hipStreamEndCapture here).
            import gc
            m = self.module
            for k, v in list(vars(m).items()):          # and no stale autograd graph stays alive
                if torch.is_tensor(v) and v.grad_fn is not None:
                    setattr(m, k, v.detach())
            gc.collect()
            for _ in range(2):                          # warm up allocator / MIOpen for exactly this op sequence
                self.optimizer.zero_grad(set_to_none=True)
                loss, _ = self.model(self._static)
                loss.mean().backward()
            del loss
            self.optimizer.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            if getattr(self, 'manual_dp', False) and getattr(self.opts, 'overlap_allreduce', True):
                g = self._graphs[key] = self._capture_split(graph)
            else:
                with torch.cuda.graph(graph, stream=self._stream):
                    loss, aux = self.model(self._static)
                    loss.mean().backward()
                # the backward pass of the capture created the .grad tensors inside THIS graph's memory pool; with more than
                # one live graph each replay has to point the parameters back at the gradients its graph writes
                grads = [(p, p.grad) for p in self.module.parameters() if p.grad is not None]
                g = self._graphs[key] = (graph, loss, aux, grads)
        if self._static is not batch:
            for k, v in batch.items():
                self._static[k].copy_(v)
        g[0].replay()
        # the replayed raster calls wrote face records into the operator's workspace behind its bookkeeping: an eager backward
        # pass still pending on this stream must not reuse "its" forward's records (soft_rasterize.py: invalidate_records)
        from ..soft_renderer import functional as _srf
        _srf.invalidate_records(self.device)
        if len(g) > 4:
            # graph-replay data parallelism with overlap: the first graph ends when the backward pass reaches the encoder's
            # layer-3 output; the gradients that exist by then (mesh, bones, heads, layer 4: ~80 % of the bytes) are all-reduced
            # on the communication stream WHILE the second graph runs the rest of the encoder's backward -- what DDP's bucket
            # hooks do in the reference (nnutils/train_utils.py:104-109, :277), which cannot live inside a capture
            from .. import parallel
            late, early, graph_b = g[4], g[5], g[6]
            t0 = self._comm_mark()
            work = parallel.allreduce_grads_async([gr for _, gr in late], average=True)
            graph_b.replay()
            t1 = self._comm_mark()
            parallel.allreduce_grads_([gr for _, gr in early], average=True)
            work()
            self._comm_done(t0, t1)
            self._dp_reduced = True
        for p, gr in g[3]:
            p.grad = gr
        return g[1], g[2]

    def _capture_split(self, graph_a):
        """Capture forward + backward as TWO graphs that share a memory pool, cut where the backward pass crosses the output of
        the encoder's third residual layer: graph A = forward + backward down to that tensor (autograd.grad with the boundary
        tensor and every parameter above it as inputs), graph B = the backward of the layers below it, seeded with the
        boundary's gradient.  Same kernels, same order, same numbers as the single graph."""
        trunk = self.module.encoder.resnet_conv
        if trunk.n_blocks < 2:
            raise ValueError('overlapped all-reduce needs an encoder trunk of at least two residual layers to cut between '
                             '(--nooverlap_allreduce for n_blocks = %d)' % trunk.n_blocks)
        trunk.boundary_after = min(2, trunk.n_blocks - 2)
        try:
            return self._capture_split_at(graph_a, trunk)
        finally:
            # the boundary is recorded during the capture only: later eager / eval forwards must not keep an activation (and,
            # in eager steps, its autograd graph) alive in the module
            trunk.boundary_after = -1
            trunk.boundary = None

    def _capture_split_at(self, graph_a, trunk):
        params = [p for p in self.module.parameters() if p.requires_grad]
        early_ids = {id(p) for p in trunk.early_parameters()}
        late = [p for p in params if id(p) not in early_ids]
        early = [p for p in params if id(p) in early_ids]
        with torch.cuda.graph(graph_a, stream=self._stream):
            loss, aux = self.model(self._static)
            xb = trunk.boundary
            ga = torch.autograd.grad(loss.mean(), [xb] + late, retain_graph=True, allow_unused=True)
        graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph_b, stream=self._stream, pool=graph_a.pool()):
            gb = torch.autograd.grad(xb, early, grad_outputs=ga[0], allow_unused=True)
        late_g = [(p, gr) for p, gr in zip(late, ga[1:]) if gr is not None]
        early_g = [(p, gr) for p, gr in zip(early, gb) if gr is not None]
        self._split_keepalive = (loss, xb)               # the autograd graph of the capture must outlive it (graph B replays it)
        return (graph_a, loss, aux, late_g + early_g, late_g, early_g, graph_b)

    def train_step(self, batch):
        """forward, backward (DDP all-reduces the gradients), clipping + NaN guard, AdamW, OneCycleLR (:274-296)."""
        m = self.module
        m.schedule_scalars()
        key = self._graph_key()
        if key is not None:
            total_loss, aux = self._graphed_forward_backward(batch, key)
        else:
            # eager step: gradients are dropped, not zeroed (zeroing would turn every parameter's gradient write into a
            # read-add-write launch of autograd's accumulation).  The caching allocator hands the same blocks back in steady
            # state, and the fused tail keeps one table per address set (_tail_table), so no rebuild / host sync follows.
            self.optimizer.zero_grad(set_to_none=not hasattr(self, '_graphs'))
            total_loss, aux = self.model(batch)
            total_loss.mean().backward()
        if getattr(self, 'manual_dp', False) and not self.__dict__.pop('_dp_reduced', False):
            from .. import parallel                        # mean of the ranks' gradients, one flat message over RCCL
            t0 = self._comm_mark()
            parallel.allreduce_grads_([p.grad for p in m.parameters() if p.grad is not None], average=True)
            self._comm_done(t0, t0)
        self.step_tail()
        return total_loss.detach(), aux

    # Optional timing of the gradient all-reduce for bench.py (`time_comm = True`): HIP events on the trainer's stream around the
    # collective(s) of a step.  comm_events holds (start, after_overlapped_compute, end) per step: end - start is the span from
    # the first collective's enqueue to the last one's completion as this stream sees it; end - after_overlapped_compute is the
    # part of it that is NOT hidden behind the second graph's replay (equal to the whole span without overlap).
    time_comm = False

    def _comm_mark(self):
        if not self.time_comm:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _comm_done(self, t0, t1):
        if t0 is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.__dict__.setdefault('comm_events', []).append((t0, t1, e))

    def comm_times_ms(self):
        """[(span_ms, exposed_ms)] of the recorded steps (synchronises); clears the record."""
        ev = self.__dict__.pop('comm_events', [])
        if ev:
            ev[-1][2].synchronize()
        return [(a.elapsed_time(c), b.elapsed_time(c)) for a, b, c in ev]

    def step_tail(self):
        """What follows backward() in the reference loop (nnutils/train_utils.py:282-296): clip the mean-shape gradient to
        norm 1 and the encoder / code-predictor gradients (jointly) to norm 10; if ANY gradient holds a NaN, zero every
        gradient -- zero, not None: AdamW still runs, so the step applies weight decay and the decayed momentum exactly as
        the reference's `optimizer.zero_grad()` (torch 1.7: in-place zeroing) followed by `optimizer.step()` does; then
        AdamW and the OneCycle schedule.  The reference tests `isnan` per parameter tensor with a host sync each (~70 per
        step); here one multi-tensor norm + one sync decides (an Inf gradient is treated like a NaN: it would turn into
        NaN inside clip_grad_norm_ anyway)."""
        if self._tail_table() is not None:
            return self._step_tail_hip()
        m = self.module
        cam_grad, grads = [], []
        for name, p in m.named_parameters():
            if p.grad is None:
                continue
            if name == 'mean_v':
                torch.nn.utils.clip_grad_norm_(p, 1.)
                self.grad_meanv_norm = p.grad.view(-1).norm(2, -1)           # after clipping, as the reference logs it
            elif 'code_predictor' in name or 'encoder' in name:
                cam_grad.append(p)
            grads.append(p.grad)
        self.grad_cam_norm = torch.nn.utils.clip_grad_norm_(cam_grad, 10.) if cam_grad else None
        self.skipped_nan = bool(grads) and not bool(torch.isfinite(torch.stack(torch._foreach_norm(grads)).sum()))
        if self.skipped_nan:
            torch._foreach_zero_(grads)
            self._skipped_eager = getattr(self, '_skipped_eager', 0) + 1
        self.optimizer.step()
        self.scheduler.step()
        # torch.optim advanced the step tensors of the parameters it stepped; the fused tail's cached tables carry their own
        # shared count (_tail_t), which would now be stale: forget them, the next fused step re-reads the optimizer's counts (and
        # takes the torch path for good if they have become non-uniform, e.g. a parameter without a gradient this step)
        if getattr(self, '_tail_shared', None) or getattr(self, '_tail_caches', None):
            self._tail_reset()

    # ---- the same tail as three multi-tensor HIP launches (lasr_tail_step, lasr_amd/csrc/tail.hip) ------------------------
    # Used on a GPU from the second step on: the first step goes through torch.optim.AdamW, which creates the optimizer state
    # (exp_avg, exp_avg_sq, step) that the kernels then update in place -- state_dict() / load_state_dict() keep working.
    @property
    def skipped_nan(self):
        v = getattr(self, '_skipped_nan', False)
        return bool(v() if callable(v) else v)

    @skipped_nan.setter
    def skipped_nan(self, v):
        self._skipped_nan = v

    def skipped_steps(self):
        """How many steps of this run found a NaN / Inf gradient (every gradient zeroed, AdamW still stepped).  One host read."""
        ctl = getattr(self, '_tail_ctl', None)
        n = int(ctl[6]) if ctl is not None else 0
        return n + getattr(self, '_skipped_eager', 0)

    def _tail_table(self):
        """Device table of (param, grad, exp_avg, exp_avg_sq, step, numel, group, clip class) rows, rebuilt when an address
        changes; None when the fused tail does not apply (CPU, --nofused_tail, optimizer state not created yet, ...)."""
        if self.device.type != 'cuda' or not getattr(self.opts, 'fused_tail', True):
            return None
        opt = self.optimizer
        names = {id(p): n for n, p in self.module.named_parameters()}
        rows, key = [], []
        for gi, group in enumerate(opt.param_groups):
            if group.get('amsgrad') or group.get('maximize'):
                return None
            for p in group['params']:
                if p.grad is None:
                    continue
                st = opt.state.get(p)
                if not st or 'exp_avg' not in st or not torch.is_tensor(st.get('step')) or not st['step'].is_cuda:
                    return None
                g = p.grad
                if p.dtype != torch.float32 or g.dtype != torch.float32 or not (p.is_contiguous() and g.is_contiguous()):
                    return None
                name = names.get(id(p), '')
                clip = 1 if name == 'mean_v' else (2 if ('code_predictor' in name or 'encoder' in name) else 0)
                rows.append((p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(),
                             st['step'].data_ptr(), p.numel(), gi, clip))
                key.append(st['step'])
        if not rows or len(opt.param_groups) > _lib.TAIL_MAX_GROUPS:
            return None
        # one table per set of addresses: with --use_graph the gradients alternate between the graphs' memory pools (plain /
        # pose-noise iterations, hypothesis switches).  What needs a host sync -- reading the optimizer's step counts -- depends on
        # the parameters and their state only and is done once per such set (`_tail_shared`); a set that differs in the GRADIENT
        # addresses alone (eager steps with zero_grad(set_to_none=True): the allocator hands out other blocks) costs one
        # asynchronous upload of the row table from pinned memory, no sync
        caches = self.__dict__.setdefault('_tail_caches', {})
        rkey = tuple(rows)
        cached = caches.get(rkey)
        if cached is not None:
            self._tail_cache = cached
            return cached if cached.get('table') is not None else None
        shared = self.__dict__.setdefault('_tail_shared', {})
        skey = tuple(r[:1] + r[2:] for r in rows)                                # everything but the gradient address
        sh = shared.get(skey)
        dev = self.device
        if sh is None:
            # One step count (_tail_t) serves every cached table, which is only right while they all cover the SAME parameters
            # (tables that differ in gradient addresses only).  A new parameter / state set -- a parameter without a gradient this
            # step, an edited optimizer state -- drops the tables of the old one: when that set comes back it re-reads the
            # optimizer's counts like this one does now (and goes to the torch path if they are no longer uniform).
            if shared:
                shared.clear()
                caches.clear()
            steps = torch.stack([t.reshape(()) for t in key]).cpu()             # the one sync per parameter / state set
            self._tail_syncs = getattr(self, '_tail_syncs', 0) + 1
            if float(steps.min()) != float(steps.max()):                         # tensors at different step counts (a parameter
                sh = shared[skey] = dict(valid=False)                            # joined later): torch path, decided once
            else:
                self._tail_t = int(steps[0])                                     # the optimizer's step count

                ch = _lib.lib().lasr_tail_chunk_elems()
                chunks = [(i, off) for i, r in enumerate(rows) for off in range(0, r[5], ch)]
                if getattr(self, '_tail_ctl', None) is None:
                    self._tail_ctl = torch.zeros(8, dtype=torch.float32, device=dev)   # ctl[6] counts the skipped (NaN) steps of the run
                sh = shared[skey] = dict(valid=True, n_chunks=len(chunks), chunks=torch.tensor(chunks, dtype=torch.int32).to(dev),
                                         partials=torch.empty(len(chunks), dtype=torch.float64, device=dev))
        if len(caches) > 8:
            caches.clear()                                                       # addresses keep changing (eager steps): bounded
        if not sh['valid']:
            self._tail_cache = caches[rkey] = dict(rows=rows, table=None)
            return None
        table = torch.tensor(rows, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
        self._tail_cache = cached = caches[rkey] = dict(rows=rows, n_chunks=sh['n_chunks'], table=table, chunks=sh['chunks'],
                                                        partials=sh['partials'], ctl=self._tail_ctl)
        return cached

    def _tail_reset(self):
        """Forget the cached tables (after the optimizer state was edited by hand)."""
        self._tail_cache = None
        self.__dict__.pop('_tail_caches', None)
        self.__dict__.pop('_tail_shared', None)

    def _step_tail_hip(self):
        import ctypes
        c = self._tail_cache
        groups = self.optimizer.param_groups
        n = len(groups)
        t = self._tail_t + 1

        def arr(kind, vals):
            return (kind * n)(*vals)
        b1, b2 = [g['betas'][0] for g in groups], [g['betas'][1] for g in groups]
        guard, st = _lib.stream_of(c['ctl'])
        with guard:
            rc = _lib.lib().lasr_tail_step(
                c['table'].data_ptr(), c['chunks'].data_ptr(), c['n_chunks'], c['partials'].data_ptr(), c['ctl'].data_ptr(), 1., 10.,
                arr(ctypes.c_float, [g['lr'] for g in groups]), arr(ctypes.c_float, b1), arr(ctypes.c_float, b2),
                arr(ctypes.c_float, [g['eps'] for g in groups]), arr(ctypes.c_float, [g['weight_decay'] for g in groups]),
                arr(ctypes.c_double, [1. - b ** t for b in b1]), arr(ctypes.c_double, [1. - b ** t for b in b2]), n, st)
        _lib.check(rc, 'lasr_tail_step')
        self._tail_t = t
        ctl = c['ctl']
        self.grad_meanv_norm, self.grad_cam_norm = ctl[3], ctl[4]                # device scalars, read when logged
        self._skipped_nan = lambda: float(ctl[2]) == 0.                          # host sync only if somebody asks
        self.scheduler.step()

    def reinit_bones(self):
        """Epoch-0 bone placement by k-means on the mean shape, rank 0 then broadcast (:243-256)."""
        opts, m = self.opts, self.module
        if opts.n_bones <= 1:
            return
        nb = opts.n_bones - 1
        if self.rank == 0:
            with torch.no_grad():
                for h in range(opts.n_hypo):
                    _, centres = kmeans(m.symmetrize(m.mean_v[h]).detach(), nb)
                    m.rest_ts.data[h * nb:(h + 1) * nb] = centres
                    m.ctl_ts.data[h * nb:(h + 1) * nb] = centres
                m.ctl_rs.data[:] = torch.tensor([0., 0., 0., 1.], device=m.ctl_rs.device)
                m.log_ctl.data[:] = 1.
        if self.distributed:
            dist.barrier()
            for t in (m.ctl_ts, m.rest_ts, m.ctl_rs, m.log_ctl):
                dist.broadcast(t.data, 0)

    def train(self):
        opts, m = self.opts, self.module
        total_steps = 0
        torch.manual_seed(8)
        self.epoch_nscore = torch.zeros(opts.n_hypo, device=self.device)
        self.model.train()
        for epoch in range(opts.num_epochs):
            m.epoch = epoch
            if epoch == 0:
                self.reinit_bones()
            m.optim_idx = int((-self.epoch_nscore).argmax())
            self.epoch_nscore[:] = 0
            for i, pair_ids in enumerate(self.dataloader):
                m.iters, m.total_steps = i, total_steps
                loss, aux = self.train_step(self.set_input(pair_ids))
                # the score that ranks the camera hypotheses skips the first 100 iterations of an epoch (pose / scale noise
                # is injected for 1 < i < 100 and the epoch starts with a transient), nnutils/train_utils.py:345-346.  An
                # epoch is padded to >= 200 iterations in the reference; shorter epochs (tests, demos) score every iteration.
                if i > 100 or len(self.dataloader) <= 101:
                    self.epoch_nscore += aux['current_nscore'].detach()
                total_steps += 1
            if self.distributed:                              # keep hypothesis selection identical on all ranks
                dist.all_reduce(self.epoch_nscore)
            skipped = self.skipped_steps()                    # steps whose gradients held a NaN and were zeroed (:289-290)
            if skipped and self.rank == 0:
                print('[lasr_amd] epoch %d: %d step(s) so far had a non-finite gradient and were taken with zeroed gradients'
                      % (epoch, skipped), file=sys.stderr)
            if self.rank == 0 and opts.checkpoint_dir:
                self.save('latest')
                if (epoch + 1) % max(1, opts.save_epoch_freq) == 0:
                    self.save(epoch + 1)
        return total_steps

    # ---- checkpoints (:363-378) ---------------------------------------------------------------
    def save(self, label):
        m = self.module
        states = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        # the ResNet-18 trunk also under the reference's key names (encoder.resnet_conv.resnet.layerN.*), so that the reference's
        # predictor / extract tools find the encoder weights in these checkpoints; aliases share storage (no extra bytes on disk)
        for k in list(states):
            ref = mesh_net.reference_resnet_key(k)
            if ref is not None:
                states[ref] = states[k]
        states['faces'] = m.faces.cpu()
        states['full_shape'] = [m.symmetrize(v).detach().cpu() for v in m.mean_v]
        states['full_tex'] = [m.symmetrize_color(t).detach().cpu() for t in m.tex]
        states['epoch_nscore'] = self.epoch_nscore.cpu() if hasattr(self, 'epoch_nscore') else None
        torch.save(states, os.path.join(self.save_dir, 'pred_net_%s.pth' % label))

    def load_network(self, network, model_path):
        """Warm start from the previous stage (train_utils.py:381-487): pick the best camera hypothesis when the new
        stage has fewer, re-mesh to exactly --n_faces when the symmetry constraint is dropped (own re-mesher, see remesh.py),
        keep the root bone's predictor rows and re-seed the part bones by k-means when the bone count changes, then load
        whatever still fits."""
        opts = self.opts
        states = torch.load(model_path, map_location='cpu')
        score_cams = -states['epoch_nscore'] if states.get('epoch_nscore') is not None else torch.zeros(1)
        n_old = len(score_cams)
        if opts.n_hypo < n_old:                                                  # select hypothesis (:387-416)
            best = int(score_cams.argmax())
            print('selecting hypothesis #%d' % best)
            for head in ('quat_predictor', 'scale_predictor'):
                w, b = ('code_predictor.%s.pred_layer.%s' % (head, x) for x in ('weight', 'bias'))
                states[w] = states[w].view(n_old, -1, states[w].shape[-1])[best]
                states[b] = states[b].view(n_old, -1)[best]
            states['mean_v'] = states['mean_v'][best:best + 1]
            states['tex'] = states['tex'][best:best + 1] if 'tex' in states else network.tex.data.cpu()
            if states['mean_v'].shape[1] < int(states['faces'].max()):           # symmetric half -> full mesh
                states['mean_v'] = states['full_shape'][best][None]
                states['tex'] = states['full_tex'][best][None]
            for name, width in (('ctl_rs', 4), ('rest_ts', 3), ('ctl_ts', 3), ('log_ctl', 3)):
                if name in states:
                    states[name] = states[name].view(n_old, -1, width)[best]

        # ---- mean shape / topology (:418-450)
        if (not opts.symmetric) and int(opts.n_faces) != states['faces'].shape[0]:
            from . import remesh
            src = states['mean_v'][0]
            if src.shape[0] < int(states['faces'].max()) + 1:
                src = states['full_shape'][0]
            v, f = remesh.remesh_exact(src.numpy(), states['faces'].numpy(), opts.n_faces)
            mean_shape, faces = torch.from_numpy(v), torch.from_numpy(f)
            tex = torch.zeros(1, mean_shape.shape[0], 3)
        elif opts.symmetric:
            mean_shape, faces, tex = None, states['faces'], states.get('tex')
        else:
            mean_shape, faces, tex = states['mean_v'][0], states['faces'], states.get('tex')
            if int(faces.max()) + 1 > mean_shape.shape[0]:
                mean_shape, tex = states['full_shape'][0], states['full_tex'][0][None]
        network.faces = faces.to(network.faces.device)
        if opts.symmetric:
            if states['mean_v'].shape == network.mean_v.shape:
                network.mean_v.data = states['mean_v'].clone()
        else:
            network.mean_v.data = mean_shape[None].repeat(opts.n_hypo, 1, 1)
        if tex is not None and tex.shape[1:] == network.mean_v.shape[1:]:
            network.tex.data = tex.expand(opts.n_hypo, -1, -1).clone()
        elif not opts.symmetric:
            network.tex.data = torch.zeros(opts.n_hypo, network.mean_v.shape[1], 3)
        for k in ('mean_v', 'tex', 'faces'):
            states.pop(k, None)

        # ---- from a rigid body / fewer bones to more bones (:455-484)
        key = 'code_predictor.depth_predictor.pred_layer.bias'
        if key in states and states[key].shape[0] != opts.n_bones:
            own = network.state_dict()
            nfeat = states['code_predictor.quat_predictor.pred_layer.weight'].shape[-1]
            for head, width, per_hypo in (('quat_predictor', 4, opts.n_hypo), ('trans_predictor', 2, 1), ('depth_predictor', 1, 1)):
                w, b = ('code_predictor.%s.pred_layer.%s' % (head, x) for x in ('weight', 'bias'))
                if states[w].shape[0] % (width * per_hypo) or own[w].shape[0] != opts.n_bones * width * per_hypo:
                    states.pop(w), states.pop(b)
                    continue
                old_w = states[w].view(per_hypo, -1, width, nfeat)[:, :1]        # the root bone of every hypothesis
                old_b = states[b].view(per_hypo, -1, width)[:, :1]
                new_w = own[w].view(per_hypo, opts.n_bones, width, nfeat).clone()
                new_b = own[b].view(per_hypo, opts.n_bones, width).clone()
                new_w[:, :1], new_b[:, :1] = old_w, old_b
                states[w], states[b] = new_w.reshape(own[w].shape), new_b.reshape(own[b].shape)
            if opts.n_bones > 1:                                                  # initialise the skin from the mean shape
                shape0 = network.symmetrize(network.mean_v.data[0]).detach().cpu()
                centres = kmeans(shape0, opts.n_bones - 1)[1] if opts.n_bones > 2 else shape0.mean(0)[None]
                states['rest_ts'] = centres.repeat(opts.n_hypo, 1)
                states['ctl_ts'] = centres.repeat(opts.n_hypo, 1)
                states['ctl_rs'] = network.ctl_rs.data.cpu()
                states['log_ctl'] = network.log_ctl.data.cpu()
        own = network.state_dict()
        for k in list(states):                                 # reference checkpoints name the trunk encoder.resnet_conv.resnet.layerN.*
            if k.startswith('encoder.resnet_conv.') and k not in own:
                mapped = mesh_net.map_resnet_key(k)
                if mapped is not None:
                    states['encoder.resnet_conv.' + mapped] = states.pop(k)
        network.load_state_dict({k: v for k, v in states.items()
                                 if k in own and torch.is_tensor(v) and own[k].shape == v.shape}, strict=False)
