"""Data-parallel CausalBGM smoke run with Bayesian nets (use_bnn=True): EGM (replicated), fit, predict.  On a 1-GPU box:
   BGM_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29537 \\
       scripts/dp_bnn_smoke.py gloo
Every rank must end with identical networks; the block-sharded predict must agree across ranks (tests compare it with the
single-process value)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.distributed as dist
backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
dev = int(os.environ.get("BGM_DEVICE", os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
dist.init_process_group(backend=backend)
from bayesgm_amd.models import CausalBGM
from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
x, y, v = Sim_Hirano_Imbens_sampler(N=1089, v_dim=30, seed=1).load_all()
params = dict(dataset="dp", output_dir="gpurun_out/dp", save_res=False, save_model=False, binary_treatment=False, use_bnn=True,
              z_dims=[1, 1, 1, 7], v_dim=30, lr_theta=1e-3, lr_z=1e-3, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
              e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True)
m = CausalBGM(params, random_seed=2, device=dev)
xs = np.linspace(0, 3, 6)
adrf0, interval0 = m.predict((x, y, v), alpha=0.05, n_mcmc=30, burn_in=30, x_values=xs, q_sd=0.5, bs=256, verbose=0)   # untrained, seeded
m.fit((x, y, v), epochs=2, epochs_per_eval=1, batch_size=32, use_egm_init=True, egm_n_iter=12, egm_batches_per_eval=6, verbose=0)
adrf, interval = m.predict((x, y, v), alpha=0.05, n_mcmc=30, burn_in=30, x_values=xs, q_sd=0.5, bs=256, verbose=0)
theta = m.engine.read(0)
t = torch.from_numpy(np.concatenate([theta, adrf.ravel(), interval.ravel()]).astype(np.float32)).cuda()
mx, mn = t.clone(), t.clone()
dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
spread = float((mx - mn).abs().max().item())
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dp_print import print_in_rank_order
print_in_rank_order(json.dumps(dict(rank=dist.get_rank(), fit_path=getattr(m, 'last_fit_path', None), spread=spread, adrf=[float(a) for a in adrf], adrf_untrained=[float(a) for a in adrf0],
                      interval_untrained=[float(a) for a in interval0.ravel()])))
assert spread == 0.0 and np.all(np.isfinite(adrf)) and np.all(np.isfinite(theta))
dist.destroy_process_group()
