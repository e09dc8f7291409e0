"""Launch-bound chains of LASR.forward as single HIP kernel pairs (include/lasr_ops.h, lasr_amd/csrc/fused.hip).

Each function replaces a run of eager elementwise ops of the reference (file:line in its docstring) and has a torch
restatement in oracle/path_oracle.py that the GPU tests compare against."""
import torch
from torch.autograd import Function

from .. import _lib


class _FlowReproject(Function):
    @staticmethod
    def forward(ctx, px, pp0, pp1, fl0, fl1):
        _lib.need_cuda(px, pp0, pp1, fl0, fl1)
        N = px.shape[0]
        P = px[0, 0].numel()
        px = px.contiguous().float()
        pp0, pp1 = pp0.contiguous().float(), pp1.contiguous().float()
        fl0, fl1 = fl0.contiguous().float(), fl1.contiguous().float()
        flow = torch.empty(px.shape[0], *px.shape[2:], 2, dtype=torch.float32, device=px.device)
        bg = torch.empty(px.shape[0], *px.shape[2:], dtype=torch.uint8, device=px.device)
        guard, st = _lib.stream_of(px)
        with guard:
            rc = _lib.lib().lasr_flow_reproject_forward(px.data_ptr(), pp0.data_ptr(), pp1.data_ptr(), fl0.data_ptr(),
                                                        fl1.data_ptr(), flow.data_ptr(), bg.data_ptr(), N, P, st)
        _lib.check(rc, 'lasr_flow_reproject_forward')
        ctx.save_for_backward(px, fl1)
        bg = bg.view(torch.bool)
        ctx.mark_non_differentiable(bg)
        return flow, bg

    @staticmethod
    def backward(ctx, gflow, _gbg):
        px, fl1 = ctx.saved_tensors
        N = px.shape[0]
        P = px[0, 0].numel()
        gflow = gflow.contiguous().float()
        gpx = torch.empty_like(px)
        gpp1 = torch.empty(N, 2, dtype=torch.float32, device=px.device)
        gfl1 = torch.empty(N, dtype=torch.float32, device=px.device)
        h = _lib.lib()
        scratch = torch.empty(h.lasr_flow_reproject_scratch_floats(N, P), dtype=torch.float32, device=px.device)
        guard, st = _lib.stream_of(px)
        with guard:
            rc = h.lasr_flow_reproject_backward(px.data_ptr(), fl1.data_ptr(), gflow.data_ptr(), gpx.data_ptr(),
                                                gpp1.data_ptr(), gfl1.data_ptr(), scratch.data_ptr(), N, P, st)
        _lib.check(rc, 'lasr_flow_reproject_backward')
        return gpx, None, gpp1, None, gfl1


def flow_reproject(px, pp0, pp1, fl0, fl1):
    """Tail of render_flow_soft_2 (/root/reference/nnutils/mesh_net.py:93-104): px [N,7,IS,IS] (position of frame t,
    of frame t', alpha), pp0/pp1 [N,2], fl0/fl1 [N] or [N,1] -> flow [N,IS,IS,2], bgmask [N,IS,IS] bool."""
    N = px.shape[0]
    return _FlowReproject.apply(px, pp0, pp1, fl0.reshape(N, -1)[:, 0], fl1.reshape(N, -1)[:, 0])
