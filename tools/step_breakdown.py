"""Where does a LASR optimisation step spend its time?  (spot3 stage-0 configuration, synchronised sections)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import optimize
from lasr_amd.nnutils import train_utils

opts = optimize.parse_flags(['--name', 'b', '--checkpoint_dir', '', '--only_mean_sym', '--nouse_gtpose', '--subdivide', '3',
                             '--n_bones', '21', '--n_hypo', '8', '--num_epochs', '5', '--batch_size', '1', '--opt_tex', 'yes',
                             '--iters_per_epoch', '40', '--nouse_graph'] + sys.argv[1:])
tr = train_utils.LASRTrainer(opts).init_training()
tr.model.train(); tr.reinit_bones()
m = tr.module
def sync(): torch.cuda.synchronize(); return time.perf_counter()
acc = {}
def add(k, dt): acc[k] = acc.get(k, 0.0) + dt
for i in range(25):
    m.iters = i
    batch = tr.set_input(tr.dataloader[i])
    t0 = sync()
    tr.optimizer.zero_grad()
    loss, aux = tr.model(batch)
    t1 = sync()
    loss.mean().backward()
    t2 = sync()
    cam = [p for n, p in m.named_parameters() if p.grad is not None and ('code_predictor' in n or 'encoder' in n)]
    torch.nn.utils.clip_grad_norm_(m.mean_v, 1.); torch.nn.utils.clip_grad_norm_(cam, 10.)
    fin = bool(torch.isfinite(torch.stack([p.grad.sum() for p in m.parameters() if p.grad is not None]).sum()))
    t3 = sync()
    tr.optimizer.step(); tr.scheduler.step()
    t4 = sync()
    if i >= 5:
        add('forward', t1 - t0); add('backward', t2 - t1); add('clip+nan', t3 - t2); add('adamw', t4 - t3)
n = 20
print({k: round(v / n * 1e3, 2) for k, v in acc.items()}, 'ms/iter; total', round(sum(acc.values()) / n * 1e3, 2))
# forward sub-sections via profiler
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for i in range(25, 28):
        m.iters = i
        l, _ = tr.model(tr.set_input(tr.dataloader[i])); l.backward()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=18, max_name_column_width=50))
