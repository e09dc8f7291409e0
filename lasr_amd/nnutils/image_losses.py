"""Per-(image, hypothesis) loss tables of LASR.forward (/root/reference/nnutils/mesh_net.py:374-447) as fused
HIP reductions: no python loop over images, no boolean-mask indexing, no host sync."""
import torch
from torch.autograd import Function

from .. import _lib


def _scratch(device, I, H, P):
    """Chunk partials + per-row counts; allocated per call because it is saved for the backward pass."""
    return torch.empty(_lib.lib().lasr_loss_scratch_floats(I, H, P), dtype=torch.float32, device=device)


class _MaskLoss(Function):
    @staticmethod
    def forward(ctx, pred, masks, occ):
        _lib.need_cuda(pred, masks, occ)
        I, H = pred.shape[:2]
        P = pred[0, 0].numel()
        pred, masks, occ = pred.contiguous().float(), masks.contiguous().float(), occ.contiguous().float()
        loss = torch.empty(I, H, dtype=torch.float32, device=pred.device)
        scratch = _scratch(pred.device, I, H, P)
        guard, st = _lib.stream_of(pred)
        with guard:
            rc = _lib.lib().lasr_mask_loss_forward(pred.data_ptr(), masks.data_ptr(), occ.data_ptr(), loss.data_ptr(),
                                                   scratch.data_ptr(), I, H, P, st)
        _lib.check(rc, 'lasr_mask_loss_forward')
        ctx.save_for_backward(pred, masks, occ, scratch)
        return loss

    @staticmethod
    def backward(ctx, g):
        pred, masks, occ, scratch = ctx.saved_tensors
        I, H = pred.shape[:2]
        P = pred[0, 0].numel()
        g = g.contiguous().float()
        gp = torch.empty_like(pred)
        guard, st = _lib.stream_of(pred)
        with guard:
            rc = _lib.lib().lasr_mask_loss_backward(pred.data_ptr(), masks.data_ptr(), occ.data_ptr(), g.data_ptr(),
                                                    scratch.data_ptr(), gp.data_ptr(), I, H, P, st)
        _lib.check(rc, 'lasr_mask_loss_backward')
        return gp, None, None


def mask_loss_table(mask_pred, masks, occ):
    """mask_pred [2B,H,IS,IS], masks/occ [2B,IS,IS] -> [2B,H] = 0.5 * mean_{occ!=0} (pred - mask)^2
    (mesh_net.py:374-388)."""
    return _MaskLoss.apply(mask_pred, masks, occ)


class _FlowLoss(Function):
    @staticmethod
    def forward(ctx, flow_rd, flow_obs, bg, occ, masks):
        _lib.need_cuda(flow_rd, flow_obs, bg, occ, masks)
        I, H = flow_rd.shape[:2]
        P = occ[0].numel()
        flow_rd = flow_rd.contiguous().float()
        flow_obs = flow_obs.contiguous().float()                  # [I, C>=2, IS, IS]
        stride = flow_obs[0].numel()
        bg8 = bg.contiguous()
        bg8 = bg8.view(torch.uint8) if bg8.dtype == torch.bool else bg8.to(torch.uint8)       # bool: same bytes, no copy
        occ, masks = occ.contiguous().float(), masks.contiguous().float()
        loss = torch.empty(I, H, dtype=torch.float32, device=flow_rd.device)
        fmap = torch.empty(flow_rd.shape[:-1], dtype=torch.float32, device=flow_rd.device)
        vis = torch.empty(flow_rd.shape[:-1], dtype=torch.uint8, device=flow_rd.device)
        scratch = _scratch(flow_rd.device, I, H, P)
        guard, st = _lib.stream_of(flow_rd)
        with guard:
            rc = _lib.lib().lasr_flow_loss_forward_vis(flow_rd.data_ptr(), flow_obs.data_ptr(), bg8.data_ptr(), occ.data_ptr(),
                                                       masks.data_ptr(), loss.data_ptr(), fmap.data_ptr(), vis.data_ptr(),
                                                       scratch.data_ptr(), I, H, P, stride, st)
        _lib.check(rc, 'lasr_flow_loss_forward_vis')
        ctx.save_for_backward(flow_rd, flow_obs, bg8, occ, masks, scratch)
        vis = vis.view(torch.bool)
        ctx.mark_non_differentiable(fmap, vis)
        return loss, fmap, vis

    @staticmethod
    def backward(ctx, g, _gmap, _gvis=None):
        flow_rd, flow_obs, bg8, occ, masks, scratch = ctx.saved_tensors
        I, H = flow_rd.shape[:2]
        P = occ[0].numel()
        g = g.contiguous().float()
        gf = torch.empty_like(flow_rd)
        guard, st = _lib.stream_of(flow_rd)
        with guard:
            rc = _lib.lib().lasr_flow_loss_backward(flow_rd.data_ptr(), flow_obs.data_ptr(), bg8.data_ptr(), occ.data_ptr(),
                                                    masks.data_ptr(), scratch.data_ptr(), g.data_ptr(), gf.data_ptr(),
                                                    I, H, P, flow_obs[0].numel(), st)
        _lib.check(rc, 'lasr_flow_loss_backward')
        return gf, None, None, None, None


def flow_loss_table(flow_rd, flow_obs, bgmask, occ, masks, with_vis=False):
    """flow_rd [2B,H,IS,IS,2], flow_obs [2B,>=2,IS,IS], bgmask [2B,H,IS,IS] bool, occ/masks [2B,IS,IS]
    -> (loss [2B,H], weighted error map [2B,H,IS,IS])   (mesh_net.py:393-413).  with_vis: also the selection mask
    (~bgmask & (occ != 0) & (masks > 0), the reference's `vis_mask`) the kernel evaluates anyway."""
    loss, fmap, vis = _FlowLoss.apply(flow_rd, flow_obs, bgmask, occ, masks)
    return (loss, fmap, vis) if with_vis else (loss, fmap)


class _TexLoss(Function):
    @staticmethod
    def forward(ctx, img_obs, img_white, rnd, fg, occ, wt):
        _lib.need_cuda(img_obs, img_white, rnd, fg, occ)
        I, H = rnd.shape[:2]
        P = occ[0].numel()
        ts = [t.contiguous().float() for t in (img_obs, img_white, rnd, fg, occ)]
        loss = torch.empty(I, H, dtype=torch.float32, device=rnd.device)
        scratch = _scratch(rnd.device, I, H, P)
        guard, st = _lib.stream_of(rnd)
        with guard:
            rc = _lib.lib().lasr_tex_loss_forward(*[t.data_ptr() for t in ts], loss.data_ptr(), scratch.data_ptr(),
                                                  float(wt), I, H, P, st)
        _lib.check(rc, 'lasr_tex_loss_forward')
        ctx.save_for_backward(*ts, scratch)
        ctx.wt = float(wt)
        return loss

    @staticmethod
    def backward(ctx, g):
        *ts, scratch = ctx.saved_tensors
        rnd, fg, occ = ts[2], ts[3], ts[4]
        I, H = rnd.shape[:2]
        P = occ[0].numel()
        g = g.contiguous().float()
        grnd, gfg = torch.empty_like(rnd), torch.empty_like(fg)
        guard, st = _lib.stream_of(rnd)
        with guard:
            rc = _lib.lib().lasr_tex_loss_backward(*[t.data_ptr() for t in ts], g.data_ptr(), scratch.data_ptr(),
                                                   grnd.data_ptr(), gfg.data_ptr(), ctx.wt, I, H, P, st)
        _lib.check(rc, 'lasr_tex_loss_backward')
        return None, None, grnd, gfg, None, None


def tex_loss_table(img_obs, img_white, texture_render, fgmask, occ, l1tex_wt=1.0):
    """img_obs/img_white [2B,3,IS,IS], texture_render [2B,H,3,IS,IS], fgmask [2B,H,IS,IS], occ [2B,IS,IS] -> [2B,H]
    = 2*wt*(mean_{occ!=0} mean_c |obs - rnd*fg| + mean_{occ!=0} mean_c |white - rnd|)   (mesh_net.py:425-441)."""
    return _TexLoss.apply(img_obs, img_white, texture_render, fgmask, occ, l1tex_wt)
