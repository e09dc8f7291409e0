import sys, faulthandler; faulthandler.enable()
sys.path.insert(0, '.')
import torch, optimize
from lasr_amd.nnutils import train_utils
extra = sys.argv[1:]
opts = optimize.parse_flags(['--name', 'b', '--checkpoint_dir', '', '--only_mean_sym', '--nouse_gtpose', '--subdivide', '2',
                             '--n_bones', '5', '--n_hypo', '2', '--num_epochs', '5', '--batch_size', '1', '--opt_tex', 'yes',
                             '--img_size', '64', '--iters_per_epoch', '10', '--use_graph'] + extra)
tr = train_utils.LASRTrainer(opts).init_training()
tr.model.train(); tr.reinit_bones()
for i in range(8):
    tr.module.iters = i
    l, _ = tr.train_step(tr.set_input(tr.dataloader[i]))
    torch.cuda.synchronize()
    print(i, float(l), flush=True)
