"""Data-parallel BGM(use_bnn=True) smoke run: fit + predict with two ranks.  On a 1-GPU box:
   BGM_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 \\
       scripts/dp_bgm_bnn_smoke.py gloo
On a multi-GPU node use `nccl` and drop BGM_DEVICE.  Prints one JSON line per rank; every rank must end with the same
generator and the same imputation."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.distributed as dist
backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
dev = int(os.environ.get("BGM_DEVICE", os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
dist.init_process_group(backend=backend)
from bayesgm_amd.models import BGM
rs = np.random.RandomState(0)
q, p, n = 4, 12, 1025                                    # shards differ by one row
data = (rs.standard_normal((n, q)) @ rs.standard_normal((q, p)) + 0.1 * rs.standard_normal((n, p))).astype(np.float32)
bp = dict(dataset="dp", output_dir="gpurun_out/dp", save_res=False, save_model=False, use_bnn=True, z_dim=q, x_dim=p,
          lr_theta=5e-3, lr_z=5e-3, g_units=[32, 32], e_units=[32, 32], dz_units=[16, 8], dx_units=[16, 8],
          kl_weight=5e-5, lr=1e-3, g_d_freq=1, alpha=0.0, gamma=0.0, bnn_mcmc_noise="frozen")
m = BGM(bp, random_seed=3, device=dev)
m.fit(data, epochs=8, epochs_per_eval=4, use_egm_init=False, verbose=0)
miss = data[:257].copy()
miss[::3, -1] = np.nan
miss[1::5, 2] = np.nan                                  # ragged missing pattern, rows sharded over the ranks, bs-blocks split
imp, interval = m.predict(miss, bs=100, n_mcmc=20, burn_in=20, step_size=0.05, num_leapfrog_steps=3)
iv = np.concatenate([np.asarray(a, np.float32).ravel() for a in interval]) if isinstance(interval, list) else interval.ravel()
from bayesgm_amd.bvn_engine import flatten_vnet
flat = np.concatenate([flatten_vnet(m.g), imp.ravel(), iv])
t = torch.from_numpy(flat).cuda()
mx, mn = t.clone(), t.clone()
dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
spread = float((mx - mn).abs().max().item())
digest = float(np.abs(imp[np.isnan(miss)]).sum())
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dp_print import print_in_rank_order
print_in_rank_order(json.dumps(dict(rank=dist.get_rank(), rows=int(m.data_z.shape[0]), history=[float(h) for h in m.history_loss], param_spread=spread,
                      imputed_digest=digest)))
assert spread == 0.0 and np.isfinite(flat).all()      # (8 epochs from random latents do not reduce the noisy MSE yet)
dist.destroy_process_group()
