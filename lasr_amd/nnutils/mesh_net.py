"""LASR model: analysis-by-synthesis forward pass on the MI355X kernels.

Mirror of the reference model API (/root/reference/nnutils/mesh_net.py:115-556 `LASR`, and the mesh container of
/root/reference/third_party/ext_nnutils/mesh_net.py:60-185 `MeshNet`): `LASR(input_shape, opts, nz_feat)`,
`forward(batch_input) -> (total_loss, aux_output)`, the same batch keys (trailing blanks included), the same
parameter names (`mean_v`, `tex`, `ctl_rs`, `rest_ts`, `ctl_ts`, `log_ctl`, `encoder.*`, `code_predictor.*`) so the
trainer's optimiser groups and checkpoints line up.  What changes is how the rendering + loss section runs:

  reference                                              here
  K-1 bmm launches + [N,K-1,V,3] temporary per LBS       one MFMA kernel (lasr_lbs_forward)
  6 elementwise kernels per pinhole_cam                  one kernel
  brute-force CUDA rasteriser with float atomics         tile-binned / face-major HIP kernels (lasr_sr_*)
  python loops over (image, hypothesis) + mask indexing  fused loss-table kernels, no host sync
  dense [V,V] Laplacian matmul, six dense [N,V,V] ARAP   CSR gathers
  dead LBS + projection for `verts_mask` (:341-345)      dropped (never rendered in the reference)

The ResNet-18 encoder and the AlexNet perceptual term are dense convolutions that stay on PyTorch/MIOpen
(SURVEY.md section 2, rows marked OUT); without network access they are randomly initialised.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import soft_renderer as sr
from .. import synth
from . import fused_ops, image_losses
from .geom_utils import obj_to_cam, obj_to_cam_both, pinhole_cam


# ----------------------------------------------------------------------------------------------
# small math helpers standing in for kornia / ext_utils (absent here; formulas are the standard ones)
# ----------------------------------------------------------------------------------------------
def quaternion_to_rotation_matrix(q):
    """(x, y, z, w) quaternion -> 3x3, normalising first (kornia 0.5.3 semantics, call sites mesh_net.py:232,250,265)."""
    if q.is_cuda:
        return fused_ops.quat_to_rotmat(q)                # one kernel instead of ~35 (and ~80 in autograd's backward)
    q = F.normalize(q, p=2, dim=-1, eps=1e-12)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    m = torch.stack([1 - (ty * y + tz * z), tx * y - tz * w, tx * z + ty * w,
                     tx * y + tz * w, 1 - (tx * x + tz * z), ty * z - tx * w,
                     tx * z - ty * w, ty * z + tx * w, 1 - (tx * x + ty * y)], -1)
    return m.reshape(q.shape[:-1] + (3, 3))


def geodesic_distance(m1, m2):
    """Rotation angle between two batches of 3x3 matrices (ext_utils/util_rot.py:27-37).  Same values; where the two
    rotations coincide (cos rounds to +-1) the reference's acos(min(cos, 1)) back-propagates 0 * inf = NaN and its trainer
    skips the step -- here that element simply has zero gradient."""
    if m1.is_cuda:
        return fused_ops.geodesic_distance(m1, m2)                             # one kernel (lasr_geodesic_forward/backward)
    m = torch.bmm(m1, m2.transpose(1, 2))
    cos = (m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2] - 1) / 2
    inside = cos.abs() < 1
    angle = torch.acos(torch.where(inside, cos, torch.zeros_like(cos)))
    return torch.where(inside, angle, torch.where(cos > 0, torch.zeros_like(cos), torch.full_like(cos, math.pi)))


def pose_noise_quat(n, t, device):
    """n noise rotations as (x, y, z, w) quaternions with the reference's law (mesh_net.py:220-232): a rotation drawn
    UNIFORMLY from SO(3) (Shoemake's construction, third_party/ext_utils/quatlib.py:22-27 `q_rnd_m`) pulled towards the
    identity by spherical interpolation with factor t (quatlib.py:29-50 `q_scale_m`): its angle -- density proportional to
    1 - cos(theta) on [0, pi] -- is multiplied by t, the axis stays uniform.  Runs on the device with torch's generator
    (the reference draws on the host with numpy); t may be a 0-dim device tensor (graph capture)."""
    u, v, w = torch.rand(3, n, device=device).unbind(0)
    v, w = 2 * math.pi * v, 2 * math.pi * w
    q = torch.stack([(1 - u).sqrt() * v.sin(), (1 - u).sqrt() * v.cos(), u.sqrt() * w.sin(), u.sqrt() * w.cos()], 1)   # wxyz
    q = torch.where(q[:, :1] < 0, -q, q)                      # shorter arc to the identity (1, 0, 0, 0)
    d = q[:, :1].clamp(max=1.0)
    t0 = torch.acos(d)
    tt = t0 * t
    s1 = torch.sin(tt) / torch.sin(t0).clamp_min(1e-12)
    s0 = torch.cos(tt) - d * s1
    ident = torch.zeros_like(q)
    ident[:, 0] = 1
    slerp = s0 * ident + s1 * q
    lin = ident + t * (q - ident)                              # d > 0.999: normalised linear interpolation (quatlib.py:38-41)
    out = torch.where(d > 0.999, F.normalize(lin, dim=1), slerp)
    return torch.cat([out[:, 1:], out[:, :1]], 1)              # -> xyzw for quaternion_to_rotation_matrix (mesh_net.py:231)


def reg_decay(curr_steps, max_steps, min_wt, max_wt):
    """Exponential decay of a regulariser weight from max_wt to min_wt (mesh_net.py:106-113)."""
    if curr_steps > max_steps:
        return min_wt
    return float(np.exp(curr_steps / float(max_steps) * (np.log(min_wt) - np.log(max_wt))) * max_wt)


def chamfer_distance(a, b):
    """Symmetric squared Chamfer distance averaged over the batch (pytorch3d.loss.chamfer_distance()[0] as used
    at mesh_net.py:503); the point sets here are the <= 35 control points."""
    if a.is_cuda:                                     # one kernel each way (lasr_chamfer_forward / backward)
        return fused_ops.chamfer(a, b).mean()
    d = (a[:, :, None] - b[:, None]).pow(2).sum(-1)
    return (d.min(2)[0].mean(1) + d.min(1)[0].mean(1)).mean()


def nearest_index(a, b):
    """For each point of a [1,V,3] the index of the nearest point of b [1,V,3] (idx1 of chamfer3D, mesh_net.py:477)."""
    if a.is_cuda:
        return fused_ops.nearest_point(a, b)[1]
    return (a[:, :, None] - b[:, None]).pow(2).sum(-1).argmin(2)


def point_mesh_face_distance(verts, faces, points):
    """mean_p min_f d^2(p, f) + mean_f min_p d^2(p, f), averaged over the batch
    (pytorch3d.loss.point_mesh_face_distance as used at mesh_net.py:470-471)."""
    return fused_ops.point_mesh_face_distance(verts, faces, points)       # HIP only; raises TypeError for CPU tensors


# ----------------------------------------------------------------------------------------------
# camera / pose regressor (dense convolutions: PyTorch/MIOpen, random init; SURVEY.md section 2 "OUT")
# ----------------------------------------------------------------------------------------------
class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class ResNetConv(nn.Module):
    """ResNet-18 trunk without the classifier (net_blocks.py:291-313 wraps torchvision's; same layer shapes)."""

    def __init__(self, n_blocks=4):
        super().__init__()
        self.n_blocks = n_blocks
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layers = nn.ModuleList()
        cin = 64
        for i, cout in enumerate((64, 128, 256, 512)):
            self.layers.append(nn.Sequential(_BasicBlock(cin, cout, 1 if i == 0 else 2), _BasicBlock(cout, cout, 1)))
            cin = cout

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        for i, layer in enumerate(self.layers[:self.n_blocks]):
            x = layer(x)
            if i == self.boundary_after:
                self.boundary = x          # the trainer cuts the captured backward here (graph-replay data parallelism); set only
                                           # while it captures (train_utils._capture_split), so no activation outlives a step
        return x

    boundary_after = -1                     # index of the layer whose output is kept in .boundary (-1: none)
    boundary = None

    def early_parameters(self):
        """Parameters whose gradients only exist after the backward pass has gone below the boundary."""
        early = list(self.conv1.parameters()) + list(self.bn1.parameters())
        for layer in self.layers[:self.boundary_after + 1]:
            early += list(layer.parameters())
        return early


def reference_resnet_key(own_key):
    """Inverse of map_resnet_key for a key of THIS model's state_dict: 'encoder.resnet_conv.layers.1.0.conv1.weight' ->
    'encoder.resnet_conv.resnet.layer2.0.conv1.weight' (the name /root/reference/third_party/ext_nnutils/net_blocks.py:291-313
    gives it), or None when the key is not a trunk tensor."""
    pre = 'encoder.resnet_conv.'
    if not own_key.startswith(pre):
        return None
    k = own_key[len(pre):]
    head = k.split('.')[0]
    if head in ('conv1', 'bn1'):
        return pre + 'resnet.' + k
    if head == 'layers':
        _, idx, rest = k.split('.', 2)
        return pre + 'resnet.layer%d.%s' % (int(idx) + 1, rest)
    return None


def map_resnet_key(key):
    """torchvision / reference names of the ResNet-18 trunk -> this module's names, or None if the key is not a trunk
    tensor.  torchvision: 'layer2.0.downsample.1.weight'; reference checkpoints: 'encoder.resnet_conv.resnet.layer2...'
    (net_blocks.py:291-313 wraps torchvision.models.resnet18 as `self.resnet`); here: 'layers.1.0.downsample.1.weight'."""
    for prefix in ('encoder.resnet_conv.resnet.', 'resnet_conv.resnet.', 'resnet.', 'module.', ''):
        if key.startswith(prefix):
            k = key[len(prefix):]
            break
    head = k.split('.')[0]
    if head in ('conv1', 'bn1'):
        return k
    if head.startswith('layer') and head[5:].isdigit() and 1 <= int(head[5:]) <= 4:
        return 'layers.%d.%s' % (int(head[5:]) - 1, k.split('.', 1)[1])
    return None                                            # fc.*, avgpool, anything else


def load_resnet18_weights(trunk, path):
    """Load a local torchvision resnet18 state_dict (or any checkpoint holding one, e.g. a reference LASR .pth) into a
    ResNetConv.  Returns the number of tensors loaded; raises if the file holds none that fit."""
    states = torch.load(path, map_location='cpu')
    states = states.get('state_dict', states) if isinstance(states, dict) else states
    own = trunk.state_dict()
    picked = {}
    for k, v in states.items():
        m = map_resnet_key(k)
        if m in own and torch.is_tensor(v) and own[m].shape == v.shape:
            picked[m] = v
    if not picked:
        raise ValueError('%s holds no ResNet-18 trunk tensors' % path)
    trunk.load_state_dict(picked, strict=False)
    return len(picked)


def _fc_stack(nc_in, nc_out, n):
    mods = []
    for _ in range(n):
        mods += [nn.Linear(nc_in, nc_out), nn.BatchNorm1d(nc_out), nn.LeakyReLU(0.2, inplace=True)]
        nc_in = nc_out
    return nn.Sequential(*mods)


class Encoder(nn.Module):
    """trunk -> conv(512->256, k4, s2) -> 2 FC layers -> nz_feat (net_blocks.py:316-339)."""

    def __init__(self, input_shape, n_blocks=4, nz_feat=100):
        super().__init__()
        self.resnet_conv = ResNetConv(n_blocks=4)
        self.enc_conv1 = nn.Sequential(nn.Conv2d(512, 256, 4, 2, 1, bias=True), nn.BatchNorm2d(256),
                                       nn.LeakyReLU(0.2, inplace=True))
        self.enc_fc = _fc_stack(256 * (input_shape[0] // 64) * (input_shape[1] // 64), nz_feat, 2)
        for m in self.enc_conv1.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, 0.02)
                m.bias.data.zero_()

    def forward(self, img):
        f = self.enc_conv1(self.resnet_conv(img))
        return self.enc_fc(f.view(img.size(0), -1))


class _Linear(nn.Module):
    def __init__(self, nz, nout):
        super().__init__()
        self.pred_layer = nn.Linear(nz, nout)


class QuatPredictor(_Linear):
    def __init__(self, nz_feat, n_bones, n_hypo):
        super().__init__(nz_feat, 4 * n_bones * n_hypo)
        self.nmesh, self.nhypo = n_bones, n_hypo

    def forward(self, feat, raw=False):
        quat = self.pred_layer(feat).view(-1, self.nhypo, self.nmesh, 4)
        bias = torch.zeros_like(quat)
        bias[:, :, 1:, 3] = 10                       # part bones start near identity (net_blocks.py:353)
        quat = F.normalize((quat + bias).view(-1, 4))
        if raw:                                      # the caller turns the quaternions into matrices itself (fused_ops.pose_chain)
            return quat
        return quaternion_to_rotation_matrix(quat).reshape(-1, 9)


class DepthPredictor(_Linear):
    def __init__(self, nz, n_bones, offset=10):
        super().__init__(nz, n_bones)
        self.offset = offset

    def forward(self, feat):
        return F.relu(self.pred_layer(feat) + self.offset) + 1e-12


class TransPredictor(_Linear):
    def __init__(self, nz, n_bones):
        super().__init__(nz, 2 * n_bones)

    def forward(self, feat):
        return self.pred_layer(feat).view(-1, 2)


class PPointPredictor(_Linear):
    def __init__(self, nz):
        super().__init__(nz, 2)

    def forward(self, feat):
        return self.pred_layer(feat)


class CodePredictor(nn.Module):
    """feat -> (scale [2B,H], trans [2B*K,2], rot [2B*H*K,9], depth [2B,K], ppoint [2B,2]) (net_blocks.py:424-450)."""

    def __init__(self, nz_feat=100, num_verts=1000, n_bones=None, n_hypo=None):
        super().__init__()
        self.offset = 20
        torch.manual_seed(0)
        self.quat_predictor = QuatPredictor(nz_feat, n_bones, n_hypo)
        self.scale_predictor = DepthPredictor(nz_feat, n_hypo, self.offset)
        self.trans_predictor = TransPredictor(nz_feat, n_bones)
        self.depth_predictor = DepthPredictor(nz_feat, n_bones, self.offset)
        self.ppoint_predictor = PPointPredictor(nz_feat)
        self.nmesh, self.nhypo = n_bones, n_hypo

    def forward(self, feat, raw_quat=False):
        scale = self.scale_predictor(feat)
        quat = self.quat_predictor(feat, raw_quat)          # raw_quat: unit quaternions [2B*H*K,4] instead of matrices
        trans = self.trans_predictor(feat) / 10.
        depth = self.depth_predictor(feat).view(-1, 1, self.nmesh)
        depth = torch.cat([depth[:, :, :1], (depth[:, :, 1:] - self.offset) / 10.], 2).view(feat.shape[0], -1)
        ppoint = self.ppoint_predictor(feat) / 10.
        return scale, trans, quat, depth, ppoint


class PerceptualDistance(nn.Module):
    """AlexNet-feature cosine distance (third_party/PerceptualSimilarity, models/networks_basic.py:42-64, called at
    mesh_net.py:442).  The pretrained weights cannot be downloaded here: random init, frozen."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        cfg = [(3, 64, 11, 4, 2), (64, 192, 5, 1, 2), (192, 384, 3, 1, 1), (384, 256, 3, 1, 1), (256, 256, 3, 1, 1)]
        self.convs = nn.ModuleList([nn.Conv2d(a, b, k, s, p) for a, b, k, s, p in cfg])
        for c in self.convs:
            c.weight.data = torch.randn(c.weight.shape, generator=g) * math.sqrt(2.0 / (c.weight[0].numel()))
            c.bias.data.zero_()
        for p_ in self.parameters():
            p_.requires_grad_(False)
        self.register_buffer('shift', torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1))
        self.register_buffer('scale', torch.tensor([.458, .448, .450]).view(1, 3, 1, 1))
        # the same normalisation for inputs given in [0, 1] that stand for 2 x - 1: ((2x - 1) - shift) / scale = x * gain + bias
        self.register_buffer('unit_gain', 2. / self.scale, persistent=False)
        self.register_buffer('unit_bias', -(1. + self.shift) / self.scale, persistent=False)

    def load_weights(self, path):
        """AlexNet convolution weights from a local state_dict: torchvision alexnet ('features.{0,3,6,8,10}.weight'), or the
        LPIPS 'alex' network of third_party/PerceptualSimilarity ('net.slice{1..5}.{0,3,6,8,10}.weight').  The reference's
        perceptual term is the unweighted cosine distance of these five feature maps (networks_basic.py:42-64)."""
        states = torch.load(path, map_location='cpu')
        states = states.get('state_dict', states) if isinstance(states, dict) else states
        index = {'0': 0, '3': 1, '6': 2, '8': 3, '10': 4}
        n = 0
        for k, v in states.items():
            parts = k.split('.')
            if len(parts) >= 2 and parts[-1] in ('weight', 'bias') and parts[-2] in index and torch.is_tensor(v):
                tgt = getattr(self.convs[index[parts[-2]]], parts[-1])
                if tgt.shape == v.shape:
                    tgt.data.copy_(v)
                    n += 1
        if n != 10:
            raise ValueError('%s: expected the 5 AlexNet convolutions (10 tensors), found %d that fit' % (path, n))
        self.pretrained = True
        return n

    def _feats(self, x, unit_range=False):
        """Feature maps of the five AlexNet convolutions.  unit_range: x is in [0, 1] and stands for 2 x - 1 (the reference maps
        its images to [-1, 1] first, mesh_net.py:436-441); ((2x - 1) - shift) / scale == x * (2 / scale) - (1 + shift) / scale is
        folded into the input normalisation instead of costing two more passes over the images."""
        if unit_range:                                         # x * (2 / scale) - (1 + shift) / scale, one pass
            x = torch.addcmul(self.unit_bias, x, self.unit_gain)
        else:
            x = (x - self.shift) / self.scale
        out = []
        for i, c in enumerate(self.convs):
            if i in (1, 2):
                x = F.max_pool2d(x, 3, 2)
            x = F.relu(c(x))
            out.append(x)
        return out

    def forward_pair(self, a, b, repeat=1, unit_range=False):
        """Distance between a[i // repeat] and b[i]: `a` holds each observed image once where the reference feeds the
        network `repeat` identical copies of it (one per hypothesis, mesh_net.py:436-441).  unit_range: see _feats."""
        feats_a, feats_b = self._feats(a, unit_range), self._feats(b, unit_range)
        if all(f.is_cuda for f in feats_b) and not any(f.requires_grad for f in feats_a):
            # all five layers: normalise + dot + spatial mean + the sum over the layers in one launch (one more in the backward)
            return fused_ops.cosine_distance_layers(feats_a, feats_b, repeat)
        d = 0
        for fa, fb in zip(feats_a, feats_b):
            if fb.is_cuda and not fa.requires_grad:
                d = d + fused_ops.cosine_distance(fa, fb, repeat)            # normalise + dot + spatial mean: one kernel
                continue
            if repeat > 1:
                fa = fa.repeat_interleave(repeat, 0)
            d = d + (1. - cos_sim(fa, fb))
        return d


def normalize_tensor(x, eps=1e-10):
    """PerceptualSimilarity/util/util.py:71-74."""
    return x / (x.pow(2).sum(1, keepdim=True).sqrt() + eps)


def cos_sim(a, b):
    """PerceptualSimilarity/util/util.py:76-83: mean over pixels of the cosine between channel vectors -> [N]."""
    return (normalize_tensor(a) * normalize_tensor(b)).sum(1).mean((1, 2))


# ----------------------------------------------------------------------------------------------
# mesh container
# ----------------------------------------------------------------------------------------------
def make_symmetric(verts, faces, idx=0):
    """Reorder a mirror-symmetric mesh as [on-plane, positive side, mirrored negatives] (the layout of
    ext_utils/mesh.py:44-87).  Returns verts, faces, num_indept, num_sym (faces keep their order: textures are
    per-vertex on this path, so the reference's face re-sorting is not needed)."""
    v = np.asarray(verts, np.float64)
    tol = 1e-6
    center = np.where(np.abs(v[:, idx]) <= tol)[0]
    right = np.where(v[:, idx] > tol)[0]
    left = np.where(v[:, idx] < -tol)[0]
    assert len(left) == len(right), 'mesh is not mirror symmetric'
    flip = np.ones(3)
    flip[idx] = -1
    d = ((v[right] * flip)[:, None] - v[left][None]).__pow__(2).sum(-1)
    match = left[d.argmin(1)]
    assert len(set(match.tolist())) == len(right) and d.min(1).max() < 1e-8, 'mesh is not mirror symmetric'
    order = np.concatenate([center, right, match])
    inv = np.empty(len(order), np.int64)
    inv[order] = np.arange(len(order))
    nv = v[order]
    nv[:len(center), idx] = 0
    nv[len(center) + len(right):] = nv[len(center):len(center) + len(right)] * flip     # exact mirror images
    return nv.astype(np.float32), inv[np.asarray(faces)], len(center), len(right)


class MeshNet(nn.Module):
    def __init__(self, input_shape, opts, nz_feat=100):
        super().__init__()
        self.opts = opts
        self.symmetric = opts.symmetric
        self.symmetric_texture = opts.symmetric_texture
        verts, faces = synth.geodesic_sphere(2 ** opts.subdivide)          # 3 -> 642 verts, 1280 faces
        num_verts = verts.shape[0]
        self.texture_type = 'vertex'
        if self.symmetric:
            verts, faces, num_indept, num_sym = make_symmetric(verts, faces, opts.symidx)
            num_sym_output = num_indept + num_sym
            self.num_output = num_verts if opts.only_mean_sym else num_sym_output
            self.num_sym, self.num_indept = num_sym, num_indept
            mean_v = torch.from_numpy(verts[:num_sym_output])
            tex = torch.normal(torch.zeros(num_sym_output, 3), 1)
            flip = torch.ones(1, 3)
            flip[0, opts.symidx] = -1
            self.register_buffer('flip', flip)
            mask = torch.ones(num_sym_output + num_sym, 3)          # vertices on the symmetry plane stay on it
            mask[:num_indept, opts.symidx] = 0
            self.register_buffer('sym_mask', mask, persistent=False)
        else:
            self.num_output = num_verts
            mean_v = torch.from_numpy(verts)
            tex = torch.normal(torch.zeros(num_verts, 3), 1)
        self.mean_v = nn.Parameter(mean_v[None].repeat(opts.n_hypo, 1, 1))
        tex = tex[None].repeat(opts.n_hypo, 1, 1)
        if opts.opt_tex == 'yes':
            self.tex = nn.Parameter(tex)
        else:
            self.register_buffer('tex', tex)
        self.register_buffer('faces', torch.from_numpy(np.asarray(faces, np.int64)))
        self.encoder = Encoder(input_shape, n_blocks=4, nz_feat=nz_feat)
        self.code_predictor = CodePredictor(nz_feat=nz_feat, num_verts=self.num_output, n_bones=opts.n_bones,
                                            n_hypo=opts.n_hypo)

    def symmetrize(self, V):
        """[..., num_indept+num_sym, 3] -> full mesh (ext_nnutils/mesh_net.py:128-149); identity if not symmetric.
        Leading dimensions (the hypotheses) are batched: one cat + one multiply for all of them."""
        if not self.symmetric:
            return V
        out = torch.cat([V, self.flip * V[..., -self.num_sym:, :]], -2)
        return out * self.sym_mask

    def symmetrize_color(self, V):
        return torch.cat([V, V[..., -self.num_sym:, :]], -2) if self.symmetric else V

    def get_mean_shape(self, local_batch_size):
        """-> mean_v [2B*H,V,3], tex [2B*H,V,3] (sigmoid), faces [2B,F,3] (ext_nnutils/mesh_net.py:171-185)."""
        n2 = 2 * local_batch_size
        key = (n2, self.faces.data_ptr(), tuple(self.faces.shape), self.faces._version)       # _version: in-place edits count too
        if getattr(self, '_faces_key', None) != key:                         # connectivity is fixed: repeat it once, not per step
            self._faces_key, self._faces_n2, self._faces_rep, self._faces_inc = key, self.faces[None].repeat(n2, 1, 1), None, None
        faces = self._faces_n2
        if self.mean_v.is_cuda:                                                # symmetrise + sigmoid + tile: one kernel
            sym = self.symmetric
            mean_v, tex = fused_ops.mean_shape(self.mean_v, self.tex, self.flip if sym else None, self.sym_mask if sym else None,
                                               n2, self.num_sym if sym else 0)
            return mean_v, tex, faces
        mean_v = self.symmetrize(self.mean_v)
        tex = self.symmetrize_color(self.tex)
        mean_v = mean_v[None].repeat(n2, 1, 1, 1).view(n2 * mean_v.shape[0], -1, 3)
        tex = tex.sigmoid()[None].repeat(n2, 1, 1, 1).view(n2 * tex.shape[0], -1, 3)      # sigmoid once per hypothesis
        return mean_v, tex, faces


# ----------------------------------------------------------------------------------------------
def render_flow_soft_2(renderer_soft, verts, faces, verts_pos0, verts_pos1, pp0, pp1, proj_cam0, proj_cam1, verts_pre=None):
    """Render the camera-space positions of frame t and t' as vertex colours on frame-t geometry and reproject both
    (mesh_net.py:75-104).  Returns flow [B*H,IS,IS,2], bgmask (bool), fgmask.  `faces` may already be repeated per
    hypothesis; `verts_pre` (the pre-transformed vertices, (verts + eye) * (1,-1,1)) may be passed when the caller has
    them already -- LASR.forward computes them once for its three render calls."""
    if faces.shape[0] != verts.shape[0]:
        n_hypo = verts.shape[0] // faces.shape[0]
        faces = faces[:, None].repeat(1, n_hypo, 1, 1).view(-1, faces.shape[1], 3)
    if verts_pre is None:
        eye = sr.functional.const_tensor(renderer_soft.transform.transformer._eye, verts.device)[None, None]
        verts_pre = (verts[:, :, :3] + eye) * sr.functional.const_tensor([1, -1, 1], verts.device)
    # The reference renders the batch [verts_pre; verts_pre] with textures [pos0; pos1]: the same geometry twice.
    # One pass with 6 attribute channels gives the same two images (and the same alpha) for half the raster work.
    px = renderer_soft.render_mesh(sr.Mesh(verts_pre, faces,
                                           textures=torch.cat([verts_pos0[:, :, :3], verts_pos1[:, :, :3]], -1),
                                           texture_type='vertex'))
    fgmask = px[:, -1]
    flow, bgmask = fused_ops.flow_reproject(px, pp0, pp1, proj_cam0[:, :1], proj_cam1[:, :1])      # (:93-104)
    return flow, bgmask, fgmask


def _plain_camera(renderer, device):
    """True when `renderer`'s stages in front of the rasteriser reduce to `vertices - eye` on the geometry and nothing on the
    attributes: constant look_at eye whose rotation is exactly the identity, orthographic with unit scale, ambient-only white
    light, no anti-aliasing -- the configuration of LASR's five renderers (mesh_net.py:132-149).  LASR.forward then forms the
    rasteriser's inputs per face corner in one launch (fused_ops.raster_faces) instead of going through sr.Mesh."""
    from ..soft_renderer.functional import cameras
    t = renderer.transform
    tr = t.transformer
    if t.camera_mode != 'look_at' or tr.perspective or tr.viewing_scale != 1 or renderer.rasterizer.anti_aliasing:
        return False
    if not isinstance(tr._eye, (list, tuple)) or len(tr._eye) != 3:
        return False
    if renderer.lighting._constant_light() != [1.0, 1.0, 1.0]:
        return False
    return cameras._const_look_at(tr._eye, [0, 0, 0], [0, 1, 0], device)[1] is None


class LASR(MeshNet):
    def __init__(self, input_shape, opts, nz_feat=100):
        super().__init__(input_shape, opts, nz_feat)
        nbm1, H = opts.n_bones - 1, opts.n_hypo
        ident = torch.tensor([[0., 0., 0., 1.]])
        self.register_buffer('rest_rs', ident.repeat(nbm1, 1))
        self.register_buffer('transg', torch.zeros(nbm1, 3))
        tensors = dict(ctl_rs=ident.repeat(H * nbm1, 1), rest_ts=torch.zeros(H * nbm1, 3),
                       ctl_ts=torch.zeros(H * nbm1, 3), log_ctl=torch.zeros(H * nbm1, 3))
        for name, t in tensors.items():
            if opts.n_bones > 1:
                setattr(self, name, nn.Parameter(t))
            else:
                self.register_buffer(name, t)
        common = dict(image_size=opts.img_size, sigma_val=1e-4, camera_mode='look_at', perspective=False,
                      light_mode='vertex', light_intensity_ambient=1., light_intensity_directionals=0.)
        # the five renderers of mesh_net.py:132-149 (renderer_soft is constructed there but never rendered)
        self.renderer_soft = sr.SoftRenderer(aggr_func_rgb='hard', **common)
        self.renderer_softflf = sr.SoftRenderer(gamma_val=1e-2, aggr_func_rgb='softmax', **common)
        self.renderer_softflb = sr.SoftRenderer(gamma_val=1e-2, aggr_func_rgb='softmax', **common)
        self.renderer_softtex = sr.SoftRenderer(gamma_val=1e-2, aggr_func_rgb='softmax', **common)
        self.renderer_softpart = sr.SoftRenderer(gamma_val=1e-4, aggr_func_rgb='softmax', **common)
        self.epoch, self.iters, self.total_steps = 0, 0, 0
        self.register_buffer('reg_factor', torch.tensor(0.5), persistent=False)
        self.register_buffer('noise_decay', torch.tensor(0.2), persistent=False)
        self.optim_idx = 0
        # criteria attached by the trainer in the reference (train_utils.py:113-123); None until then
        self.triangle_loss_fn_sr = self.arap_loss_fn = self.flatten_loss = self.ptex_loss = None

    def __setattr__(self, name, value):
        # a new connectivity (load_network, re-meshing) may land on the old tensor's address with the old shape: drop the
        # repeated copies forward() caches, whatever the key says
        if name == 'faces':
            self.__dict__.pop('_faces_key', None)
            self.__dict__['_faces_n2'] = self.__dict__['_faces_rep'] = self.__dict__['_faces_inc'] = None
        super().__setattr__(name, value)

    def schedule_scalars(self):
        """Refresh the schedule-dependent scalars of forward() from (epoch, iters): the regulariser decay (:106-113, :449)
        and the pose-noise amplitude (:221).  Called by the trainer before every step."""
        self.reg_factor.fill_(reg_decay(self.epoch, self.opts.num_epochs, 0.05, 0.5))
        self.noise_decay.fill_(0.2 * (1e-4) ** (self.iters / 100))

    def _skinning(self, pred_v, n2):
        """GMM skinning weights (mesh_net.py:264-271): softmax_k(-10 * sum_d exp(log_ctl) * ((ctl_ts - v) R(ctl_rs))_d^2)."""
        opts = self.opts
        H = opts.n_hypo
        if pred_v.is_cuda:
            return fused_ops.skin_weights(self.ctl_ts, self.ctl_rs, self.log_ctl, pred_v.view(n2, H, -1, 3)[0])[..., None]
        v0 = pred_v.view(n2, H, -1, 3)[0, :, None].detach()                       # H,1,V,3
        dis = self.ctl_ts.view(H, -1, 1, 3) - v0                                  # H,J,V,3
        dis = dis.matmul(quaternion_to_rotation_matrix(self.ctl_rs).view(H, -1, 3, 3))
        dis = self.log_ctl.exp().view(H, -1, 1, 3) * dis.pow(2)
        return (-10 * dis.sum(3)).softmax(1)[:, :, :, None]                        # H,J,V,1

    @property
    def cam_export(self):
        """Root camera per (image, hypothesis) of the last training-mode forward: R, T and the intrinsics back in the UNcropped
        image, as extract.py / nnutils/predictor.py:188-189 of the reference write them."""
        Rmat, Tmat, scale, ppoint, cams, pp, n2, H, K, IS = self._cam_src
        half = IS / 2.
        with torch.no_grad():
            return dict(R=Rmat.reshape(n2 * H, K, 3, 3)[:, 0], T=Tmat.reshape(n2 * H, K, 3)[:, 0],
                        focal=(scale / cams[:, :1] * half)[:, :, None].repeat(1, 1, 2).reshape(n2 * H, 2),
                        pp=((ppoint + 1) * half / cams[:, :1] + pp)[:, None].repeat(1, H, 1).reshape(n2 * H, 2))

    def forward(self, batch_input):
        opts = self.opts
        if not self.training:
            feat = self.encoder(batch_input)
            return self.code_predictor(feat)
        B = batch_input['input_imgs  '].shape[0] // 2
        n2, H, K, IS = 2 * B, opts.n_hypo, opts.n_bones, opts.img_size
        bi = {k: v.view(B, 2, -1).permute(1, 0, 2).reshape(v.shape) for k, v in batch_input.items()}   # :155-156
        self.input_imgs, self.imgs, self.masks = bi['input_imgs  '], bi['imgs        '], bi['masks       ']
        self.cams, self.depth_gt, self.flow = bi['cams        '], bi['depth_gt    '], bi['flow        ']
        self.ddts_barrier, self.pp, self.occ = bi['ddts_barrier'], bi['pp          '], bi['occ         ']
        self.oriimg_shape, self.frameid, self.dataid = bi['oriimg_shape'], bi['frameid'], bi['dataid']

        pred_v, tex, faces = self.get_mean_shape(B)                              # [N,V,3], [N,V,3], [2B,F,3]
        for m in self.modules():                                                 # BN always in eval mode (:190-195)
            if isinstance(m, nn.modules.batchnorm._BatchNorm):                   # incl. SyncBatchNorm (class-name match in the reference)
                m.eval()
        noise_now = opts.noise and self.epoch > 0 and 1 < self.iters < 100
        # the pose chain in one launch each way (fused_ops.pose_chain) whenever nothing sits between its stages: no pose noise,
        # no ground-truth poses
        chain = self.input_imgs.is_cuda and not noise_now and not opts.use_gtpose
        scale, trans, quat, depth, ppoint = self.code_predictor(self.encoder(self.input_imgs), raw_quat=chain)
        pair_angle = None
        if chain:
            scale, ppoint, Rmat, Tmat, trans, depth, pair_angle, proj = fused_ops.pose_chain(
                self.cams, self.pp, scale, depth, ppoint, quat, trans, self.rest_ts, self.ctl_ts, H, K, IS)
            skin = None
            if K > 1:
                skin_h = self._skinning(pred_v, n2)
                skin = skin_h.repeat(n2, 1, 1, 1)
                self.joints_proj, self.ctl_proj = proj.split([K - 1, K - 1], 1)
        else:
            # intrinsics bookkeeping (:204-217): crop scale into focal length / root depth, frame t' shares frame t's principal point
            scale, depth, ppoint = fused_ops.intrinsics(self.cams, self.pp, scale, depth, ppoint, IS)
            depth = depth.reshape(-1, 1)

            quat = quat.view(-1, 9)
            if noise_now:                                                        # pose / scale noise (:220-235)
                decay = self.noise_decay                          # 0.2 * 1e-4 ** (iters / 100), device scalar (schedule_scalars)
                noise = pose_noise_quat(quat.shape[0], decay, quat.device)
                quat = quat.view(-1, 3, 3).matmul(quaternion_to_rotation_matrix(noise)).view(-1, 9)
                scale = scale * (decay * torch.randn_like(scale) * opts.rscale).exp()
            depth = depth.view(n2, 1, K, 1).repeat(1, H, 1, 1).view(-1, 1)
            trans = trans.view(n2, 1, K, 2).repeat(1, H, 1, 1).view(-1, 2)

            if opts.use_gtpose:                                                      # (:240-253)
                quat_pred, scale_pred, trans_pred = quat.clone(), scale.clone(), trans.clone()
                ppoint_pred, depth_pred = ppoint.clone(), depth.clone()
                scale = 10 * self.cams[:, :1]
                trans = self.cams[:, 1:3]
                quat = quaternion_to_rotation_matrix(torch.cat((self.cams[:, 4:], self.cams[:, 3:4]), -1)).view(-1, 9)
                depth = self.depth_gt[:]
                halforisize = 0.5 * IS / self.cams[:, :1]
                ppoint = (0.5 * self.oriimg_shape - self.pp[:]) / halforisize - 1

            # ---- rigid + articulated transforms (:259-289): Rmat = predicted matrix transposed, Tmat = (trans, depth); bones
            # k >= 1 rotate about their joint (T' = -R c + T + c with c = rest_ts) and are transposed back -- one kernel (row a3)
            # (without ground-truth poses the camera term is the rotation distance between the two frames' matrices, :514-516: it rides on
            # the fix-up's launches)
            pair_angle = None
            if opts.use_gtpose:
                Rmat, Tmat = fused_ops.bone_fixup(quat, trans, depth, self.rest_ts, H, K)
            else:
                Rmat, Tmat, pair_angle = fused_ops.bone_fixup(quat, trans, depth, self.rest_ts, H, K, pair_angle=True)
            skin = None
            if K > 1:
                skin_h = self._skinning(pred_v, n2)
                skin = skin_h.repeat(n2, 1, 1, 1)
                # joints and control points through the same transforms and the projection, one launch each way (the reference: two
                # obj_to_cam calls with an identity skin, :285-288, and pinhole_cam); only rest_ts / ctl_ts receive gradient
                if Rmat.is_cuda:
                    proj = fused_ops.project_points(self.rest_ts, self.ctl_ts, Rmat, Tmat, ppoint, scale, H, K)
                else:
                    eye = torch.eye(K - 1, device=Rmat.device)[None, :, :, None]
                    eye = torch.cat([eye, eye], 2)                                   # [1, K-1, 2(K-1), 1]
                    pts = torch.cat([self.rest_ts.view(H, K - 1, 3), self.ctl_ts.view(H, K - 1, 3)], 1).repeat(n2, 1, 1)
                    jc = obj_to_cam(pts, Rmat.detach(), Tmat[:, None].detach(), K, H, eye)
                    proj = pinhole_cam(torch.cat([jc, torch.ones_like(jc[:, :, :1])], -1), ppoint.detach(), scale.detach())
                self.joints_proj, self.ctl_proj = proj.split([K - 1, K - 1], 1)
        # ---- 1) flow rendering (:298-335); deform_v (:291) is the same blend before the body transform: one launch for both
        verts_cam, self.deform_v = obj_to_cam_both(pred_v, Rmat, Tmat[:, None, :], K, H, skin)
        self.verts_cam = verts_cam.detach()                                      # per-frame shape in camera space (export)
        # what cam_export needs (views, no kernels: the export arithmetic runs only when extract.py asks for it)
        self._cam_src = (Rmat.detach(), Tmat.detach(), scale.detach(), ppoint.detach(), self.cams, self.pp, n2, H, K, IS)
        # pinhole projection (:302), near / far from the projected depth range (:304-311, kept on the device), the raster-space
        # vertices (:81-82, :354-355) and the nine vertex attributes of the merged render: one launch (fused_ops.raster_inputs)
        N, BH = n2 * H, B * H
        pp_all = ppoint[:, None].repeat(1, H, 1).view(N, 2)
        sc_all = scale.reshape(N)
        r_tex = self.renderer_softtex
        eye = r_tex.transform.transformer._eye
        faces_rep = self._faces_rep                                             # reset by get_mean_shape when the key changes
        if faces_rep is None:
            faces_rep = self._faces_rep = faces[:, None].repeat(1, H, 1, 1).view(-1, faces.shape[1], 3)
            self._faces_inc = None
        r_tex.rasterizer.background_color = [1, 1, 1, 0, 0, 0, 0, 0, 0]
        part_now = K > 1 and self.iters == 0
        fast = verts_cam.is_cuda and not part_now and _plain_camera(r_tex, verts_cam.device)
        if fast:
            # the same values per FACE CORNER in one launch each way: projection, eye shift of the camera stage, both face gathers
            # and the near / far fold (fused_ops.raster_faces); the connectivity is shared by the batch, its incidence lists are
            # built once per face tensor
            if getattr(self, '_faces_inc', None) is None:
                self._faces_inc = (faces[:1].contiguous(),) + fused_ops.face_incidence(faces[:1], verts_cam.shape[1])
            f1, inc_ptr, inc = self._faces_inc
            fv, fattr, near_far = fused_ops.raster_faces(verts_cam, tex, pp_all, sc_all, eye, f1, (inc_ptr, inc))
        else:
            verts_pre, attrs, near_far = fused_ops.raster_inputs(verts_cam, tex, pp_all, sc_all, eye)
        near, far = near_far[0], near_far[1]                                     # views of one 2-float device tensor
        for r in (self.renderer_softflf, self.renderer_softflb, self.renderer_softtex):
            r.rasterizer.near, r.rasterizer.far = near, far
            if opts.sigval != 1e-4:
                r.rasterizer.sigma_val = opts.sigval
        if opts.sigval != 1e-4:
            self.renderer_soft.rasterizer.sigma_val = opts.sigval

        # ---- 1) + 3) flow and texture / silhouette rendering (:298-363) in ONE pass.  The reference makes three render calls:
        # flow t -> t' (frame-t geometry, attributes = camera-space positions of frames t and t'), flow t' -> t, and the
        # texture render of all frames, whose geometry it recomputes from a clone of the same Rmat (verts_tex == verts_fl).
        # All three rasterise the same 2B*H meshes with the same settings, so one 9-attribute pass (colour, own position, the
        # other frame's position; background white / black / black) yields the same images -- channels are blended
        # independently -- for one distance / sigmoid / depth evaluation per fragment instead of two.
        if fast:
            rz = r_tex.rasterizer
            px = sr.functional.soft_rasterize(fv, fattr, rz.image_size, rz.background_color, rz.near, rz.far, rz.fill_back, rz.eps,
                                              rz.sigma_val, rz.dist_func, rz.dist_eps, rz.gamma_val, rz.aggr_func_rgb,
                                              rz.aggr_func_alpha, 'vertex')
        else:
            px = r_tex.render_mesh(sr.Mesh(verts_pre, faces_rep, textures=attrs, texture_type='vertex'))
        self.texture_render, alpha = px[:, :3], px[:, 9]                         # views of the wide render
        self.mask_pred = alpha
        # ---- reprojection + the three image-loss tables (+ the perceptual net's input pair) in one pass over the render and
        # one pass back (fused_ops.render_tables; mesh_net.py:87-104, :374-441)
        want_pair = self.ptex_loss is not None
        # (the observed object on black | on white, :364-366, is formed by the same pass and comes back as its last output)
        rt = fused_ops.render_tables_imgs(px, self.masks, self.occ, self.flow, self.imgs, pp_all, sc_all, opts.l1tex_wt, want_pair)
        obspair = rt[-1]
        self.mask_loss_sub, self.flow_rd_loss_sub, tex_l1, self.flow_rd, self.bgmask, flow_map, vis = rt[:7]
        self.flow_rd_map, self.vis_mask = flow_map.view(n2, H, IS, IS), vis.view(n2, H, IS, IS)
        self.flow_fw, self.flow_bw = self.flow_rd[:BH], self.flow_rd[BH:]         # the reference's per-direction attributes (views)
        self.bgmask_fw, self.bgmask_bw = self.bgmask[:BH], self.bgmask[BH:]
        self.fgmask_flowf, self.fgmask_flowb = self.mask_pred[:BH], self.mask_pred[BH:]
        if K > 1 and self.iters == 0:                                            # part rendering, logging only (:368-370)
            with torch.no_grad():
                cmap = torch.tensor(synth.label_palette(K - 1), dtype=torch.float32, device=tex.device)
                skin_colors = (skin_h[self.optim_idx] * cmap[:, None]).sum(0) / 256.
                self.part_render = self.renderer_softpart.render_mesh(
                    sr.Mesh(verts_pre.view(n2, H, -1, 3)[:1, self.optim_idx].detach(), faces[:1],
                            textures=skin_colors[None], texture_type='vertex'))[:, :3]

        # ---- losses (:374-530).  The reference accumulates `total_loss += w * x.mean()` term by term; here every term is
        # queued as (tensor, weight, group) and ONE kernel forms the weighted means, the per-loss scalars LASR logs (group
        # totals) and the total in the same order (fused_ops.weighted_mean_sum).
        G_MASK, G_FLOW, G_TEX, G_TRI, G_SYM, G_LMOTION, G_ARAP, G_BONESYM, G_CAM, G_AUX = range(10)
        terms = []
        # 1) silhouette (:374-390), 2) flow (:393-416), 3) texture L1 (:419-441): tables of render_tables above
        terms.append((self.mask_loss_sub, 1., G_MASK))
        terms.append((self.flow_rd_loss_sub, 1., G_FLOW))
        tmp = tex_l1
        if self.ptex_loss is not None:
            # rndpair = (render * alpha | render) comes out of render_tables; the observed side is the same image for all H
            # hypotheses: its features are computed once per image
            # (the [0,1] -> [-1,1] map of :436-441 is folded into the network's input normalisation: unit_range)
            percept = self.ptex_loss.forward_pair(obspair, rt[7], repeat=H, unit_range=True)
            tmp = torch.add(tmp, percept.view(2, -1).sum(0).view(n2, H), alpha=0.005)
        self.texture_loss_sub = 0.25 * tmp
        terms.append((self.texture_loss_sub, 1., G_TEX))

        # 4) shape smoothness (:449-459)
        # a device scalar the trainer refreshes (schedule_scalars), so that one captured graph serves every epoch
        factor = 1 if H > 1 else self.reg_factor
        from . import loss_utils
        stock = (pred_v.is_cuda and type(self.triangle_loss_fn_sr) is loss_utils.LaplacianLoss and not self.triangle_loss_fn_sr.average
                 and type(self.flatten_loss) is loss_utils.FlattenLoss and not self.flatten_loss.average
                 and (K == 1 or type(self.arap_loss_fn) is loss_utils.ARAPLoss))
        arap_l = cham_l = None
        if stock:
            # Laplacian + flatten on the mean shape and ARAP between the two frames' shapes: one launch each way
            # (fused_ops.mesh_regularisers), values and gradients bit-identical to the three criteria called one by one
            if K > 1:
                dv0, dv1 = self.deform_v.reshape(2, B * H, -1, 3).unbind(0)
            else:
                dv0 = dv1 = pred_v.new_empty(0, pred_v.shape[1], 3)
            # (the bones' symmetric Chamfer term, :500-503, rides on the same launches)
            cham_pair = None
            if K > 1 and opts.symmetric_loss:
                ca = self.ctl_ts.view(H, -1, 3)
                cham_pair = (ca, ca * sr.functional.const_tensor([-1, 1, 1], ca.device))
            res = fused_ops.mesh_regularisers(pred_v, dv0, dv1, self.triangle_loss_fn_sr, self.flatten_loss,
                                              self.arap_loss_fn if K > 1 else self.triangle_loss_fn_sr, cham_pair)
            lap_l, flat_l, arap_l = res[:3]
            cham_l = res[3] if cham_pair is not None else None
        else:
            lap_l, flat_l = self.triangle_loss_fn_sr(pred_v), self.flatten_loss(pred_v)
        tri = lap_l * (factor * (0.005 * (4 ** opts.subdivide) / 64.))
        tri = tri + flat_l * (factor * (5e-4 * (2 ** opts.subdivide / 8.0)))
        self.triangle_loss_sub = tri.view(n2, H)
        terms.append((self.triangle_loss_sub, 1., G_TRI))
        if (not opts.symmetric) and opts.symmetric_loss:                          # symmetry (:461-478)
            pa = pred_v.view(n2, H, -1, 3)[0]
            pb = pa * sr.functional.const_tensor([-1, 1, 1], pa.device)
            terms.append((point_mesh_face_distance(pa, self.faces, pb).view(1), 1., G_SYM))
            terms.append((point_mesh_face_distance(pb, self.faces, pa).view(1), 1., G_SYM))
            if opts.opt_tex == 'yes':
                p1 = pred_v[:1].detach()
                idx1 = nearest_index(p1, p1 * sr.functional.const_tensor([-1, 1, 1], p1.device))
                terms.append(((self.tex[0][idx1[0]].detach() - self.tex[0]).abs(), 1e-3, G_SYM))
        # 5) deformation (:481-497)
        if K > 1:
            self.lmotion_loss_sub = (self.deform_v - pred_v).norm(2, -1).mean(-1).view(n2, H)
            if torch.is_tensor(factor):                                          # H == 1: the schedule's device scalar (else 1)
                self.lmotion_loss_sub = factor * self.lmotion_loss_sub
            terms.append((self.lmotion_loss_sub, 1., G_LMOTION))
            if arap_l is None:
                dv0, dv1 = self.deform_v.reshape(2, B * H, -1, 3).unbind(0)
                arap_l = self.arap_loss_fn(dv0, dv1)
            terms.append((arap_l, (4 ** opts.subdivide) / 64., G_ARAP))
            if opts.symmetric_loss:                                              # bone symmetry (:500-503)
                if cham_l is None:
                    ca = self.ctl_ts.view(H, -1, 3)
                    cham_l = fused_ops.chamfer(ca, ca * sr.functional.const_tensor([-1, 1, 1], ca.device))
                terms.append((cham_l, 0.1, G_BONESYM))
        # 7) camera (:506-522)
        if opts.use_gtpose:
            terms.append((geodesic_distance(quat.view(-1, 3, 3), quat_pred.view(-1, 3, 3)), 0.2, G_CAM))
            for pred, gt in ((scale_pred, scale), (trans_pred, trans), (depth_pred, depth), (ppoint_pred, ppoint)):
                terms.append(((pred - gt).abs(), 0.2, G_CAM))
        else:
            if pair_angle is None:
                q0, q1 = quat.view(2, -1, 3, 3).unbind(0)
                pair_angle = geodesic_distance(q0, q1)
            terms.append((pair_angle, 0.001, G_CAM))
            if K > 1:
                t0, t1 = trans.view(2, B * H, K, 2).unbind(0)
                d0, d1 = depth.view(2, B * H, K, 1).unbind(0)
                terms.append(((t0 - t1)[:, 1:].abs(), 0.01, G_CAM))
                terms.append(((d0 - d1)[:, 1:].abs(), 0.01, G_CAM))
        # 8) aux (:524-530): keep the root in front of the camera; joints and control points inside the silhouette
        terms.append((F.relu(2 - Tmat.view(-1, 1, K, 3)[:, :, :1, -1]), 0.02, G_AUX))
        if K > 1:
            barrier = self.ddts_barrier.repeat(1, H, 1, 1).view(-1, 1, IS, IS)
            # 100 * (0.1 * mean(joints) + 0.1 * mean(control points)) = 20 * mean over both sets (equal sizes), one sampling call
            both = F.grid_sample(barrier, proj[:, :, :2].reshape(-1, 2 * (K - 1), 1, 2), padding_mode='border', align_corners=False)
            terms.append((both, 20., G_AUX))
        total, sums = fused_ops.weighted_mean_sum(terms, 10)
        self.total_loss = total
        self.mask_loss, self.flow_rd_loss, self.texture_loss = sums[G_MASK], sums[G_FLOW], sums[G_TEX]
        self.triangle_loss, self.cam_loss = sums[G_TRI], sums[G_CAM]
        if K > 1:
            self.lmotion_loss, self.arap_loss = sums[G_LMOTION], sums[G_ARAP]

        aux = dict(flow_rd_map=self.flow_rd_map, flow_rd=self.flow_rd, vis_mask=self.vis_mask, mask_pred=self.mask_pred,
                   total_loss=self.total_loss, mask_loss=self.mask_loss, texture_loss=self.texture_loss,
                   flow_rd_loss=self.flow_rd_loss, triangle_loss=self.triangle_loss)
        if K > 1:
            aux['lmotion_loss'] = self.lmotion_loss
            aux['ctl_proj'] = self.ctl_proj
        # per-hypothesis scores (:532-541): mean over the images of each table, and their sum (texture + flow + mask)
        per_hypo = torch.stack([self.mask_loss_sub, self.flow_rd_loss_sub, self.texture_loss_sub], 0).detach().mean(1)   # [3,H]
        aux['current_nscore'] = per_hypo.sum(0)
        if H > 1:
            for h in range(H):                                                   # views of one [3,H] table
                aux['mask_hypo_%d' % h] = per_hypo[0, h]
                aux['flow_hypo_%d' % h] = per_hypo[1, h]
                aux['tex_hypo_%d' % h] = per_hypo[2, h]
        aux['texture_render'] = self.texture_render
        if hasattr(self, 'part_render'):
            aux['part_render'] = self.part_render
        return self.total_loss, aux
