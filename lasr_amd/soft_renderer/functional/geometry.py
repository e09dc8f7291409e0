"""Per-face gathers and vertex normals.

Reference behaviour: soft_renderer/functional/face_vertices.py:4-22 and vertex_normals.py:4-37
(under /root/reference/third_party/softras/).  Same values, but the batch is flattened with one
index_select / index_add_ instead of int32 offset arithmetic on the face tensor.
"""
import torch
import torch.nn.functional as F


# Connectivity that outlives a call (LASR's repeated face tensor, a Mesh rendered every iteration): the backward's vertex-centric
# sums then run over a CSR incidence structure built once (nnutils/fused_ops.face_incidence) instead of scanning the face tensor in
# every call.  The cache is keyed by the CALLER's tensor object (a weak reference: an entry dies with its tensor, so a recycled
# address can never be mistaken for it -- and an int32 face tensor is recognised although its int64 copy is new in every call) and
# checked against its storage address, shape, dtype and version counter.  The structure is built the second time a tensor is seen
# -- a face tensor made for one call costs nothing.
# Edits the version counter does not see (`faces.data[...] = ...`, `set_`) leave a stale structure behind: call
# invalidate_incidence(faces) after such an edit (in-place operations on the tensor itself are tracked).
_INC_CACHE = {}
_INC_CACHE_MAX = 16
_INC_CLOCK = [0]


def _inc_signature(faces, num_vertices):
    return (faces.data_ptr(), tuple(faces.shape), faces.dtype, faces._version, num_vertices)


def invalidate_incidence(faces=None):
    """Forget the cached incidence structure of `faces` (or of every tensor): needed only after an edit that bypasses the
    tensor's version counter."""
    if faces is None:
        _INC_CACHE.clear()
    else:
        _INC_CACHE.pop(id(faces), None)


def _incidence_of(faces, num_vertices, faces_long=None):
    """faces: the tensor the caller passed (the cache key); faces_long: its contiguous int64 form, if already made."""
    import weakref
    key = id(faces)
    ent = _INC_CACHE.get(key)
    _INC_CLOCK[0] += 1
    if ent is not None and (ent[0]() is not faces or ent[1] != _inc_signature(faces, num_vertices)):
        ent = None
    if ent is None:
        if len(_INC_CACHE) >= _INC_CACHE_MAX:
            # entries of dead tensors go first; then the least recently seen tensor whose structure was never built (nothing holds
            # it).  A structure that was handed out may sit in a captured HIP graph, which needs it for as long as it needs the face
            # tensor itself -- and that tensor's entry keeps the structure alive: such entries stay
            for k in [k for k, e in _INC_CACHE.items() if e[0]() is None]:
                _INC_CACHE.pop(k, None)
            if len(_INC_CACHE) >= _INC_CACHE_MAX:
                idle = [(e[3], k) for k, e in _INC_CACHE.items() if e[2] is None]
                if not idle:
                    return None                          # every slot holds a live structure: this tensor keeps the scanning kernel
                _INC_CACHE.pop(min(idle)[1], None)
        _INC_CACHE[key] = [weakref.ref(faces), _inc_signature(faces, num_vertices), None, _INC_CLOCK[0]]
        return None
    ent[3] = _INC_CLOCK[0]
    if ent[2] is None:
        if faces.is_cuda and torch.cuda.is_current_stream_capturing():
            return None                                  # (built eagerly, outside a capture: argsort allocates)
        from ...nnutils import fused_ops
        fl = faces_long if faces_long is not None else faces.contiguous().long()
        ent[2] = fused_ops.face_incidence(fl if fl.dim() == 3 else fl[None], num_vertices)      # [F,3]: one shared mesh
    return ent[2]


class _FaceGather(torch.autograd.Function):
    """lasr_face_gather_* (include/lasr_ops.h)."""

    @staticmethod
    def forward(ctx, attr, faces):
        from ... import _lib
        attr = attr.contiguous()
        faces_key = faces
        faces = faces.contiguous().long()
        N, V, C = attr.shape
        F_ = faces.shape[1]
        out = torch.empty(N, F_, 3, C, dtype=torch.float32, device=attr.device)
        guard, st = _lib.stream_of(attr)
        with guard:
            rc = _lib.lib().lasr_face_gather_forward(attr.data_ptr(), faces.data_ptr(), out.data_ptr(), N, V, F_, C, st)
        _lib.check(rc, 'lasr_face_gather_forward')
        ctx.save_for_backward(faces)
        ctx.dims = (N, V, F_, C)
        ctx.inc = _incidence_of(faces_key, V, faces)
        return out

    @staticmethod
    def backward(ctx, g):
        from ... import _lib
        faces, = ctx.saved_tensors
        N, V, F_, C = ctx.dims
        g = g.contiguous().float()
        ga = torch.empty(N, V, C, dtype=torch.float32, device=g.device)
        guard, st = _lib.stream_of(g)
        with guard:
            if ctx.inc is not None:                     # connectivity seen before: the vertex-centric sums without the scan (same bits)
                inc_ptr, inc = ctx.inc
                rc = _lib.lib().lasr_face_gather_backward_csr(g.data_ptr(), inc_ptr.data_ptr(), inc.data_ptr(), 0, ga.data_ptr(),
                                                              N, V, F_, C, st)
            else:
                rc = _lib.lib().lasr_face_gather_backward(g.data_ptr(), faces.data_ptr(), ga.data_ptr(), N, V, F_, C, st)
        _lib.check(rc, 'lasr_face_gather_backward')
        return ga, None


def _flat_index(faces, num_vertices):
    bs = faces.shape[0]
    offs = torch.arange(bs, device=faces.device, dtype=torch.long) * num_vertices
    return (faces.long() + offs[:, None, None]).reshape(-1)


def face_vertices(vertices, faces):
    """[B,V,C] per-vertex attributes + [B,F,3] indices -> [B,F,3,C] per-face attributes."""
    if vertices.ndimension() != 3 or faces.ndimension() != 3:
        raise AssertionError('vertices and faces must be 3-dimensional')
    if vertices.shape[0] != faces.shape[0] or faces.shape[2] != 3:
        raise AssertionError('batch sizes must agree and faces must be [B,F,3]')
    if vertices.is_cuda and vertices.dtype == torch.float32:
        return _FaceGather.apply(vertices, faces)                   # one kernel each way; ordered backward sum
    bs, nv, ch = vertices.shape
    nf = faces.shape[1]
    flat = vertices.reshape(bs * nv, ch)
    return flat.index_select(0, _flat_index(faces, nv)).reshape(bs, nf, 3, ch)


def vertex_normals(vertices, faces):
    """Area-weighted vertex normals, unit length (eps 1e-6), [B,V,3]."""
    if vertices.shape[2] != 3:
        raise AssertionError('vertices must be [B,V,3]')
    bs, nv = vertices.shape[:2]
    idx = _flat_index(faces, nv).reshape(-1, 3)                      # [B*F,3] into the flattened vertex list
    tri = vertices.reshape(bs * nv, 3)[idx]                          # [B*F,3,3]
    p0, p1, p2 = tri[:, 0], tri[:, 1], tri[:, 2]
    acc = torch.zeros(bs * nv, 3, dtype=vertices.dtype, device=vertices.device)
    # the reference accumulates corner 1, then 2, then 0 (vertex_normals.py:27-32); keep that order so
    # the float sums match
    acc.index_add_(0, idx[:, 1], torch.cross(p2 - p1, p0 - p1, dim=1))
    acc.index_add_(0, idx[:, 2], torch.cross(p0 - p2, p1 - p2, dim=1))
    acc.index_add_(0, idx[:, 0], torch.cross(p1 - p0, p2 - p0, dim=1))
    return F.normalize(acc, eps=1e-6, dim=1).reshape(bs, nv, 3)


def surface_normals(face_verts):
    """Unit face normals from [B,F,3,3] (mesh.py:104-110 of the reference)."""
    v10 = face_verts[:, :, 0] - face_verts[:, :, 1]
    v12 = face_verts[:, :, 2] - face_verts[:, :, 1]
    return F.normalize(torch.cross(v12, v10, dim=2), p=2, dim=2, eps=1e-6)
