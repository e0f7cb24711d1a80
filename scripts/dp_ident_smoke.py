"""Two-rank IdentifiableCausalBGM.predict (rows sharded, U drawn by rank 0, ADRF / ITE reductions).  On a 1-GPU box:
   BGM_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 \\
       scripts/dp_ident_smoke.py gloo"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.distributed as dist
backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
dev = int(os.environ.get("BGM_DEVICE", os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
dist.init_process_group(backend=backend)
from bayesgm_amd.models import IdentifiableCausalBGM
from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler


def params(binary):
    return dict(dataset="dpi", output_dir="gpurun_out/dpi", save_res=False, save_model=False, binary_treatment=binary, use_bnn=False,
                z_dims=[1, 1, 1, 7], v_dim=50, lr_theta=1e-3, lr_z=1e-3, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
                e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True, n_segments=6)


x, y, v = Sim_Hirano_Imbens_sampler(N=1205, v_dim=50, seed=1).load_all()
out = dict(rank=dist.get_rank())
m = IdentifiableCausalBGM(params(False), random_seed=2, device=dev)
np.random.seed(5 + 100 * dist.get_rank())              # the ranks' host generators differ: rank 0's segments must win
adrf, interval = m.predict((x, y, v), alpha=0.05, n_mcmc=40, burn_in=40, x_values=np.linspace(0, 3, 6), q_sd=0.5, verbose=0)
out.update(adrf=[float(a) for a in adrf], interval=[float(a) for a in interval.ravel()], acc=m.last_acceptance_rate)
mb = IdentifiableCausalBGM(params(True), random_seed=3, device=dev)
xb = (x > np.median(x)).astype(np.float32)
np.random.seed(6 + 100 * dist.get_rank())
ite, iv = mb.predict((xb, y, v), alpha=0.05, n_mcmc=40, burn_in=40, q_sd=0.5, verbose=0)
out.update(ite_head=[float(a) for a in ite[:5]], ite_tail=[float(a) for a in ite[-5:]], ite_sum=float(ite.sum()), iv_sum=float(iv.sum()))
# adaptive proposal scale (q_sd <= 0): ONE acceptance window over all rows -- its count is all-reduced, every rank adapts identically
ma = IdentifiableCausalBGM(params(False), random_seed=2, device=dev)
np.random.seed(5 + 100 * dist.get_rank())
adrf_a, _ = ma.predict((x, y, v), alpha=0.05, n_mcmc=20, burn_in=160, x_values=np.linspace(0, 3, 6), q_sd=-1.0, verbose=0)
out.update(adrf_adaptive=[float(a) for a in adrf_a], acc_adaptive=ma.last_acceptance_rate)
# data-parallel fit: rows, segments and latents sharded; the fused g | f | h gradient and the prior net's gradient all-reduced per step
mf = IdentifiableCausalBGM(params(False), random_seed=4, device=dev)
np.random.seed(11)                                     # fit draws the segments and the permutations from the shared host stream
mf.fit((x, y, v), batch_size=32, epochs=2, epochs_per_eval=2, use_egm_init=True, egm_n_iter=12, egm_batches_per_eval=6, verbose=0)
flat = np.concatenate([mf.nets["g"][0][0].ravel(), mf.nets["f"][-1][1], mf.nets["h"][1][0].ravel(), mf._prior_theta.cpu().numpy().ravel()]).astype(np.float32)
t = torch.from_numpy(flat).cuda()
mx, mn = t.clone(), t.clone()
dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
out.update(fit_spread=float((mx - mn).abs().max().item()), fit_finite=bool(np.all(np.isfinite(flat))),
           fit_loss=[h["loss_postrior_z"] for h in mf.fit_history], fit_mse_v=[h.get("mse_v") for h in mf.fit_history if "mse_v" in h])
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dp_print import print_in_rank_order
print_in_rank_order(json.dumps(out))
assert out["fit_spread"] == 0.0 and out["fit_finite"]
dist.destroy_process_group()
