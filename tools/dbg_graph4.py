import sys, subprocess
if len(sys.argv) == 1:
    for w in ['fwd', 'fwdbwd_noeager', 'fwdbwd']:
        r = subprocess.run([sys.executable, __file__, w], capture_output=True, text=True)
        print(w, 'rc', r.returncode, r.stdout.strip().split('\n')[-1][:100], flush=True)
    sys.exit(0)
sys.path.insert(0, '.')
import torch, optimize
from lasr_amd.nnutils import train_utils
w = sys.argv[1]
opts = optimize.parse_flags(['--name', 'b', '--checkpoint_dir', '', '--only_mean_sym', '--nouse_gtpose', '--subdivide', '2',
                             '--n_bones', '5', '--n_hypo', '2', '--num_epochs', '5', '--batch_size', '1', '--opt_tex', 'yes',
                             '--img_size', '64', '--iters_per_epoch', '10', '--noperceptual'])
tr = train_utils.LASRTrainer(opts).init_training()
tr.model.train(); tr.reinit_bones()
m = tr.module
m.iters = 5
static = tr.set_input(tr.dataloader[0])
if w == 'fwdbwd':
    tr.train_step(tr.set_input(tr.dataloader[1]))     # an eager step on the default stream first
    for k, v in list(vars(m).items()):
        if torch.is_tensor(v) and v.grad_fn is not None: setattr(m, k, v.detach())
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        tr.optimizer.zero_grad(set_to_none=True)
        loss, _ = tr.model(static)
        if w != 'fwd': loss.backward()
torch.cuda.current_stream().wait_stream(side)
tr.optimizer.zero_grad(set_to_none=True)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    loss, aux = tr.model(static)
    if w != 'fwd': loss.backward()
gr.replay(); torch.cuda.synchronize()
print('ok', float(loss))
