"""Texture atlas -> per-face surface textures (reference: soft_renderer.cuda.load_textures, used by
/root/reference/third_party/softras/soft_renderer/functional/load_obj.py:60-93 when a textured .obj is loaded)."""
import torch

from ... import _lib


def load_textures(image, faces_uv, texture_res, is_update=None):
    """image [H,W,3] float32 (row 0 = first row sampled at v = 0), faces_uv [F,3,2] in [0,1] -> textures [F,R*R,3].
    is_update [F] int32: faces with 0 keep zeros (the reference leaves their slice of the pre-zeroed tensor untouched)."""
    _lib.need_cuda(image, faces_uv)
    image, faces_uv = image.contiguous().float(), faces_uv.contiguous().float()
    F, R = faces_uv.shape[0], int(texture_res)
    H, W = image.shape[:2]
    out = torch.zeros(F, R * R, 3, dtype=torch.float32, device=image.device)
    upd = is_update.to(device=image.device, dtype=torch.int32).contiguous() if is_update is not None else None
    guard, st = _lib.stream_of(image)
    with guard:
        rc = _lib.lib().lasr_load_textures(image.data_ptr(), faces_uv.data_ptr(), upd.data_ptr() if upd is not None else None,
                                           out.data_ptr(), F, R, H, W, st)
    _lib.check(rc, 'lasr_load_textures')
    return out
