"""One-launch batch staging (lasr_amd/dataloader/packed.py, lasr_gather_rows): the packed table must hand the model exactly
the dictionary the reference's set_input builds (nnutils/train_utils.py:164-180: per key the frame-t block and the frame-t'
block, interleaved pair-major), from a persistent buffer."""
import numpy as np
import pytest
import torch

from lasr_amd.dataloader.packed import PackedTable


def _rows(n_pairs, seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = {'imgs        ': (2, 3, 8, 8), 'masks       ': (2, 8, 8), 'cams        ': (2, 7), 'pp          ': (2, 2),
              'frameid': (2,), 'odd': (2, 5)}                      # 'odd': a segment whose length is not a multiple of 4
    return [{k: torch.randn(*s, generator=g) for k, s in shapes.items()} for _ in range(n_pairs)]


def _reference_batch(rows, ids):
    """train_utils.py:179-180 applied to the frame-major blocks :164-178 would build."""
    B = len(ids)
    out = {}
    for k in rows[0]:
        t0 = torch.stack([rows[i][k][0] for i in ids])             # frame t block
        t1 = torch.stack([rows[i][k][1] for i in ids])             # frame t' block
        v = torch.cat([t0, t1], 0)
        out[k] = v.view(2, B, -1).permute(1, 0, 2).reshape(v.shape)
    return out


def test_packed_table_layout_on_cpu():
    rows = _rows(5)
    tab = PackedTable(rows, 'cpu')
    assert tab.table.shape[0] == 5 and tab.W % 4 == 0
    for ids in ([3], [0, 4, 4], [2, 1]):
        got = tab.gather(torch.tensor(ids))
        want = _reference_batch(rows, ids)
        assert getattr(got, 'persistent', False)
        for k in want:
            assert got[k].shape == want[k].shape and torch.equal(got[k], want[k]), k
    a = tab.gather(torch.tensor([1, 2]))
    b = tab.gather(torch.tensor([3, 0]))
    assert a is b and a['cams        '].data_ptr() == b['cams        '].data_ptr()      # the same views, refilled
    # the aliasing contract: the default hands out the persistent views; clone=True gives a batch that survives the next gather
    keep = tab.gather(torch.tensor([1, 2]), clone=True)
    nxt = tab.gather(torch.tensor([3, 0]))
    want = _reference_batch(rows, [1, 2])
    assert not getattr(keep, 'persistent', False) and keep is not nxt
    for k in want:
        assert torch.equal(keep[k], want[k]) and keep[k].data_ptr() != nxt[k].data_ptr(), k


@pytest.mark.gpu
def test_gather_rows_kernel_equals_the_torch_gather(cuda):
    rows = _rows(7, seed=1)
    tab = PackedTable(rows, cuda)
    for ids in ([6], [0, 3, 3, 5], [2, 1]):
        got = tab.gather(torch.tensor(ids, device=cuda))
        want = _reference_batch(rows, ids)
        for k in want:
            assert torch.equal(got[k].cpu(), want[k]), k
    # an id outside the table is clamped, never read out of bounds
    got = tab.gather(torch.tensor([99], device=cuda))
    assert torch.equal(got['odd'].cpu(), _reference_batch(rows, [6])['odd'])


@pytest.mark.gpu
def test_synthetic_sequence_batches_come_from_one_launch(cuda):
    from lasr_amd import synth_data
    seq = synth_data.SyntheticSequence(cuda, 32, n_frames=3, nu=2)
    for ids in ([0, 2], [1, 1]):
        want = seq._rows(ids)                                       # torch assembly (stack / cat / permute)
        got = seq.batch(ids)
        assert set(got) == set(want)
        for k in want:
            assert got[k].shape == want[k].shape and torch.equal(got[k], want[k]), k
    ids = torch.tensor([2, 0], device=cuda)                          # the trainer keeps an epoch's ids on the device
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        seq.batch(ids)
        torch.cuda.synchronize()
    kernels = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    assert len(kernels) == 1 and 'gather_rows' in kernels[0], kernels
