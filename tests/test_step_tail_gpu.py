"""The optimisation-step tail as HIP kernels (lasr_tail_step) against the torch path it replaces
and against the reference's loop body (/root/reference/nnutils/train_utils.py:282-296) written out statement by statement:
same clipping, same NaN behaviour, same AdamW update, same optimizer state -- over several steps of a OneCycle schedule."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import optimize                                                    # noqa: E402
from lasr_amd.nnutils import train_utils                            # noqa: E402


def make(tmp_path, cuda, fused):
    flags = ['--name', 't', '--checkpoint_dir', str(tmp_path), '--img_size', '64', '--subdivide', '1', '--n_bones', '3',
             '--n_hypo', '2', '--batch_size', '1', '--opt_tex', 'yes', '--iters_per_epoch', '4', '--noperceptual', '--nouse_graph',
             '--dataname', 'synthetic'] + ([] if fused else ['--nofused_tail'])
    opts = optimize.parse_flags(flags)
    opts.local_rank = cuda.index or 0
    torch.manual_seed(0)
    return train_utils.LASRTrainer(opts).init_training()


def set_grads(tr, seed, scale, nan_in=None):
    g = torch.Generator().manual_seed(seed)
    for name, p in tr.module.named_parameters():
        if p.requires_grad:
            grad = (scale * torch.randn(p.shape, generator=g)).to(p.device)
            if nan_in == name:
                grad.view(-1)[3] = float('nan')
            if p.grad is None:
                p.grad = grad
            else:
                p.grad.copy_(grad)                                  # keep the gradient buffers (and the kernel's table) in place


def test_fused_tail_equals_the_torch_tail_step_by_step(tmp_path, cuda):
    a = make(tmp_path / 'a', cuda, fused=True)
    b = make(tmp_path / 'b', cuda, fused=False)
    b.module.load_state_dict(a.module.state_dict())
    plan = [(1, 5.0, None), (2, 5.0, None), (3, 1e-3, None), (4, 5.0, 'ctl_ts'), (5, 0.3, None), (6, 5.0, 'encoder.enc_fc.0.weight'),
            (7, 2.0, None)]
    for i, (seed, scale, nan_in) in enumerate(plan):
        for tr in (a, b):
            set_grads(tr, seed, scale, nan_in)
            tr.step_tail()
        if i >= 1:                                                   # first step: torch creates the optimizer state; then the
            assert a._tail_table() is not None and a._tail_t == i + 1                  # kernels count the steps themselves
        assert getattr(b, '_tail_cache', None) is None
        assert a.skipped_nan == b.skipped_nan == (nan_in is not None)
        if nan_in is None:
            assert abs(float(a.grad_meanv_norm) - float(b.grad_meanv_norm)) <= 1e-5 * max(1.0, float(b.grad_meanv_norm))
            assert abs(float(a.grad_cam_norm) - float(b.grad_cam_norm)) <= 1e-5 * float(b.grad_cam_norm)
        for (name, p), q in zip(a.module.named_parameters(), b.module.parameters()):
            d = float((p - q).abs().max())
            assert d <= 2e-6 * max(1.0, float(q.abs().max())), (i, name, d)
            if nan_in is None:                                      # .grad afterwards: clipped values, as the reference leaves them
                assert float((p.grad - q.grad).abs().max()) <= 1e-5 * max(1e-6, float(q.grad.abs().max())), (i, name)
            else:
                assert not p.grad.any() and not q.grad.any()
    # the optimizer state the kernels maintained is torch's own: same tensors, same step counts, loadable
    sa, sb = a.optimizer.state_dict(), b.optimizer.state_dict()
    for k in sb['state']:
        for field in ('exp_avg', 'exp_avg_sq'):
            x, y = sa['state'][k][field], sb['state'][k][field]
            # (the clip coefficient comes from a norm summed in another order: ~1e-6 relative, twice that in the second moment)
            assert float((x - y).abs().max()) <= 2e-5 * float(y.abs().max()) + 1e-20, (k, field)
        assert float(sa['state'][k]['step']) == float(sb['state'][k]['step']) == len(plan)
    b.optimizer.load_state_dict(sa)


def test_fused_tail_against_the_reference_loop_body(tmp_path, cuda):
    tr = make(tmp_path, cuda, fused=True)
    set_grads(tr, 1, 5.0)
    tr.step_tail()                                                   # step 1 (torch): creates the state
    set_grads(tr, 2, 5.0)                                            # mean_v norm >> 1, encoder norm >> 10
    model = copy.deepcopy(tr.module)
    special = ('mean_v', 'tex', 'ctl_rs', 'rest_ts', 'ctl_ts', 'log_ctl')
    groups = [{'params': [p for n, p in model.named_parameters() if n not in special]}]
    groups += [{'params': [getattr(model, n)]} for n in special]
    opt = torch.optim.AdamW(groups, lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-4, fused=True)
    opt.load_state_dict(copy.deepcopy(tr.optimizer.state_dict()))   # same moments, same step, same current learning rates
    for p_new, p_old in zip(model.parameters(), tr.module.parameters()):
        p_new.grad = p_old.grad.clone()
    # the reference loop body, statement by statement (train_utils.py:282-294)
    cam_grad = []
    for name, p in model.named_parameters():
        if 'mean_v' == name and p.grad is not None:
            torch.nn.utils.clip_grad_norm_(p, 1.)
        elif p.grad is not None and ('code_predictor' in name or 'encoder' in name):
            cam_grad.append(p)
        if (p.grad is not None) and (torch.isnan(p.grad).sum() > 0):
            opt.zero_grad(set_to_none=False)
    torch.nn.utils.clip_grad_norm_(cam_grad, 10.)
    opt.step()
    assert tr._tail_table() is not None
    tr.step_tail()                                                   # step 2: the HIP kernels
    assert abs(float(tr.grad_meanv_norm) - 1.0) < 1e-4 and float(tr.grad_cam_norm) > 10
    for (name, p), q in zip(tr.module.named_parameters(), model.parameters()):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max())), name
    np.testing.assert_allclose(float(torch.cat([p.grad.view(-1) for n, p in tr.module.named_parameters()
                                                if 'encoder' in n or 'code_predictor' in n]).norm()), 10.0, rtol=1e-4)


def test_fused_tail_steps_aside_when_step_counts_differ(tmp_path, cuda):
    # a tensor whose optimizer state is at another step count (e.g. a parameter that joined later) needs its own bias
    # correction: the trainer then keeps torch.optim.AdamW for every step, decided once (no per-step host sync)
    tr = make(tmp_path, cuda, fused=True)
    set_grads(tr, 1, 1.0)
    tr.step_tail()
    assert tr._tail_table() is not None
    tr.optimizer.state[tr.module.mean_v]['step'] += 3
    tr._tail_reset()                                                 # force a rebuild of the table (new step counts)
    set_grads(tr, 2, 1.0)
    before = tr.module.tex.detach().clone()
    tr.step_tail()
    assert tr._tail_table() is None and tr._tail_cache['table'] is None
    assert float((tr.module.tex - before).abs().max()) > 0           # the torch path stepped
    assert float(tr.optimizer.state[tr.module.tex]['step']) == 2 and float(tr.optimizer.state[tr.module.mean_v]['step']) == 5


def test_huge_finite_gradient_is_not_mistaken_for_nan_and_skips_are_counted(tmp_path, cuda):
    # nnutils/train_utils.py:289 tests isnan: a finite gradient of 1e20 (its square overflows fp32) must be clipped, not zeroed;
    # the sums of squares run in double.  Skipped steps are counted on the device (ctl[6]) and reported by skipped_steps().
    a = make(tmp_path / 'a', cuda, fused=True)
    set_grads(a, 1, 1.0)
    a.step_tail()                                                     # torch path creates the optimizer state
    set_grads(a, 2, 1.0)
    with torch.no_grad():
        a.module.mean_v.grad.view(-1)[0] = 1e20
    before = a.module.mean_v.detach().clone()
    a.step_tail()
    assert a._tail_table() is not None
    assert a.skipped_nan is False
    assert abs(float(a.grad_meanv_norm) - 1.0) <= 1e-4                 # clipped to norm 1 (clip_grad_norm_ semantics)
    assert torch.isfinite(a.module.mean_v).all() and not torch.equal(a.module.mean_v, before)
    assert a.skipped_steps() == 0
    set_grads(a, 3, 1.0, nan_in='ctl_ts')
    a.step_tail()
    set_grads(a, 4, 1.0, nan_in='tex')
    a.step_tail()
    assert a.skipped_nan is True and a.skipped_steps() == 2


def test_eager_steps_do_not_sync_for_the_tail_table(tmp_path, cuda):
    # --nouse_graph: zero_grad(set_to_none=True) reallocates the gradients every step and the allocator may hand out other
    # blocks each time.  The host sync of the table (reading the optimizer's step counts) belongs to the parameter / state set
    # and happens once; new gradient addresses cost an asynchronous upload only, and the table cache stays bounded
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_lasr_forward_gpu import make_trainer
    a = make_trainer(tmp_path, iters_per_epoch=12)                   # eager (use_graph=False), fused tail on by default
    a.model.train()
    a.reinit_bones()
    before = [p.detach().clone() for p in a.module.parameters()]
    for i in range(12):
        a.module.iters = i + 1
        a.train_step(a.set_input(a.dataloader[i]))
        c = getattr(a, '_tail_cache', None)
        if i >= 1:                                                   # (the first step is torch's: it creates the optimizer state)
            assert c.get('table') is not None and a._tail_t == i + 1
            assert torch.equal(c['table'].cpu(), torch.tensor(c['rows'], dtype=torch.int64))
    assert a._tail_syncs == 1 and len(a._tail_shared) == 1 and len(a._tail_caches) <= 9
    assert a.skipped_steps() == 0
    moved = [float((p - q).abs().max()) for p, q in zip(a.module.parameters(), before) if p.requires_grad]
    assert all(np.isfinite(moved)) and max(moved) > 0
    st = a.optimizer.state[a.module.mean_v]
    assert float(st['step']) == 12
