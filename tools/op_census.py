"""Which source lines of the LASR step launch the device kernels?  (eager, spot3 stage-0 configuration)

    python tools/op_census.py [optimize.py flags]

Counts device kernels and their time per innermost repository source line (forward) and per autograd node (backward),
so that the many-small-kernels tail of the step can be fused where it is largest.
"""
import collections, os, sys
import torch
from torch.profiler import profile, ProfilerActivity
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optimize
from lasr_amd.nnutils import train_utils

opts = optimize.parse_flags(['--name', 'b', '--checkpoint_dir', '', '--only_mean_sym', '--nouse_gtpose', '--subdivide', '3',
                             '--n_bones', '21', '--n_hypo', '8', '--num_epochs', '5', '--batch_size', '1', '--opt_tex', 'yes',
                             '--iters_per_epoch', '40', '--nouse_graph'] + sys.argv[1:])
tr = train_utils.LASRTrainer(opts).init_training()
tr.model.train(); tr.reinit_bones()
# section markers (monkeypatched record_function ranges; with_stack yields nothing on this build)
from torch.profiler import record_function
from lasr_amd.nnutils import mesh_net, image_losses, geom_utils
import lasr_amd.soft_renderer as sr


def mark(obj, name, label=None):
    fn = getattr(obj, name)
    def wrapped(*a, **k):
        with record_function('SEC:' + (label or name)):
            return fn(*a, **k)
    setattr(obj, name, wrapped)


m = tr.module
for name in ('mask_loss_table', 'flow_loss_table', 'tex_loss_table'):
    mark(image_losses, name)
mark(mesh_net, 'render_flow_soft_2'); mark(mesh_net, 'obj_to_cam'); mark(mesh_net, 'obj_to_cam_both', 'obj_to_cam'); mark(mesh_net, 'pinhole_cam')
mark(mesh_net, 'geodesic_distance'); mark(mesh_net, 'chamfer_distance'); mark(mesh_net, 'point_mesh_face_distance')
mark(m, '_skinning'); mark(m, 'get_mean_shape'); mark(m.encoder, 'forward', 'encoder'); mark(m.code_predictor, 'forward', 'code_predictor')
if m.ptex_loss is not None:
    mark(m.ptex_loss, 'forward_pair', 'perceptual')
mark(m.triangle_loss_fn_sr, 'forward', 'laplacian'); mark(m.flatten_loss, 'forward', 'flatten_loss'); mark(m.arap_loss_fn, 'forward', 'arap')
mark(m.renderer_softtex, 'render_mesh', 'render_mesh(tex)')
mark(tr.optimizer, 'step', 'optimizer.step'); mark(torch.nn.utils, 'clip_grad_norm_')
mark(torch.nn.functional, 'grid_sample')
for i in range(6):
    tr.module.iters = i + 1
    tr.train_step(tr.set_input(tr.dataloader[i]))
torch.cuda.synchronize()
NIT = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(6, 6 + NIT):
        tr.module.iters = i + 1
        tr.train_step(tr.set_input(tr.dataloader[i]))
    torch.cuda.synchronize()
def section_of(e):
    p = e
    while p is not None:
        if p.name.startswith('SEC:'):
            return p.name[4:]
        p = p.cpu_parent
    return None


# forward ops carry a sequence number that their autograd node repeats in the backward pass
seq2sec = {}
for e in prof.events():
    if e.sequence_nr is not None and e.sequence_nr >= 0 and not e.name.startswith('autograd::engine'):
        sec = section_of(e)
        if sec is not None and 'Backward' not in e.name:
            seq2sec.setdefault(e.sequence_nr, sec)
by = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    ks = getattr(e, 'kernels', None)
    if not ks:
        continue
    where = section_of(e)
    if where is None:
        p = e
        while p is not None:
            if p.name.startswith('autograd::engine::evaluate_function'):
                where = 'bwd of ' + seq2sec.get(p.sequence_nr, 'forward (other) [%s]' % p.name.split(': ')[-1][:40])
                break
            p = p.cpu_parent
    if where is None:
        where = 'forward (other): ' + e.name[:50]
    by[where][0] += len(ks)
    by[where][1] += sum(k.duration for k in ks)
tot_n = sum(v[0] for v in by.values()); tot_t = sum(v[1] for v in by.values())
print('kernels/iter %.0f, device time/iter %.2f ms' % (tot_n / NIT, tot_t / NIT / 1e3))
for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:90]:
    print('%6.1f kernels %8.1f us  %s' % (v[0] / NIT, v[1] / NIT, k))
