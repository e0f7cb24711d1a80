"""Data-parallel BGM.fit smoke run.  On a 1-GPU box:
   BGM_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \\
       scripts/dp_bgm_fit_smoke.py gloo
On a multi-GPU node use `nccl` and drop BGM_DEVICE.  Prints one JSON line per rank; rank 0 also checks that every rank
ended with the same generator."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.distributed as dist
backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
dev = int(os.environ.get("BGM_DEVICE", os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
dist.init_process_group(backend=backend)
from bayesgm_amd.models import BGM
from bayesgm_amd.datasets import simulate_z_hetero
X, Y = simulate_z_hetero(n=2001, k=3, d=19, seed=42)
data = np.c_[X, Y].astype(np.float32)
bp = dict(dataset="dp", output_dir="gpurun_out/dp", save_res=False, save_model=False, use_bnn=False, z_dim=10, x_dim=20,
          lr_theta=2e-3, lr_z=2e-3, g_units=[64] * 5, e_units=[64] * 5, dz_units=[64, 32, 8], dx_units=[64, 32, 8],
          kl_weight=5e-5, lr=1e-3, g_d_freq=1, use_z_rec=True, alpha=0.0, gamma=0.0)
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
# "egm": replicated EGM warm start (device-drawn reparameterisation noise) before the data-parallel fit;
# "egm_noseed": the same with random_seed=None (rank 0's seed is broadcast, bayesgm_amd.parallel.shared_seed)
m = BGM(bp, random_seed=None if mode == "egm_noseed" else 3, device=dev)
m.fit(data, epochs=6, epochs_per_eval=3, use_egm_init=mode.startswith("egm"), egm_n_iter=30, egm_batches_per_eval=15, verbose=0)
miss = data[:257].copy()
miss[::3, -1] = np.nan
miss[1::5, 2] = np.nan                                  # ragged missing pattern, rows sharded over the ranks
imp, interval = m.predict(miss, n_mcmc=20, burn_in=20)
iv = np.concatenate([np.asarray(a, np.float32).ravel() for a in interval]) if isinstance(interval, list) else interval.ravel()
flat = np.concatenate([m.g["bn"]["gamma"], m.g["bn"]["mean"], m.g["trunk"][0][0].ravel(), m.g["mean"][1], imp.ravel(), iv])
t = torch.from_numpy(flat).cuda()
mx, mn = t.clone(), t.clone()
dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
spread = float((mx - mn).abs().max().item())
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dp_print import print_in_rank_order
print_in_rank_order(json.dumps(dict(rank=dist.get_rank(), rows=int(m.data_z.shape[0]), history=[float(h) for h in m.history_loss], param_spread=spread)))
assert spread == 0.0 and m.history_loss[-1] < m.history_loss[0]
dist.destroy_process_group()
