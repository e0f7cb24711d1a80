"""Two-rank IdentifiableCausalBGM(use_bnn=True): data-parallel fit (rows, segments, latents sharded; the g | h | f gradient and the data
part of the Bayesian prior net's gradient all-reduced per step) and predict (the panel is ONE block: every rank runs it whole, rank 0's
segment draw is everybody's).  On a 1-GPU box:
   BGM_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29538 \\
       scripts/dp_ident_bnn_smoke.py gloo"""
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
prm = dict(dataset="dpib", output_dir="gpurun_out/dpib", save_res=False, save_model=False, binary_treatment=False, use_bnn=True,
           z_dims=[1, 1, 1, 7], v_dim=50, lr_theta=1e-3, lr_z=1e-3, g_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
           e_units=[64] * 5, dz_units=[64, 32, 8], kl_weight=1e-4, lr=2e-4, g_d_freq=5, use_z_rec=True, n_segments=6)
x, y, v = Sim_Hirano_Imbens_sampler(N=1205, v_dim=50, seed=1).load_all()
m = IdentifiableCausalBGM(prm, random_seed=4, device=dev)
# untrained, seeded predicts (fixed and adaptive proposal scale): the ONE block of the panel is sharded by rows over the ranks (round 6) and
# must give what a single process gives -- compared in tests/test_gpu_identifiable_bnn.py
np.random.seed(5 + 100 * dist.get_rank())              # the ranks' host generators differ: rank 0's segments must win
adrf_u, interval_u = m.predict((x, y, v), alpha=0.05, n_mcmc=20, burn_in=20, x_values=np.linspace(0, 3, 5), q_sd=0.5, verbose=0)
adrf_a, _ = m.predict((x, y, v), alpha=0.05, n_mcmc=20, burn_in=120, x_values=np.linspace(0, 3, 5), q_sd=-1.0, verbose=0)
q_sd_a = [float(a) for a in np.asarray(m.last_q_sd).ravel()]      # the block's adapted proposal scale (acceptance counts summed over the ranks)
sharded = bool(m.engine.serves_block_shares())
np.random.seed(11)                     # fit draws the segments and the permutations from the shared host stream
m.fit((x, y, v), batch_size=32, epochs=2, epochs_per_eval=2, use_egm_init=True, egm_n_iter=12, egm_batches_per_eval=6, verbose=0)
np.random.seed(5 + 100 * dist.get_rank())              # the ranks' host generators differ: rank 0's segments must win
adrf, interval = m.predict((x, y, v), alpha=0.05, n_mcmc=20, burn_in=20, x_values=np.linspace(0, 3, 5), q_sd=0.5, verbose=0)
flat = np.concatenate([m.nets["g"]["layers"][0][0].ravel(), m.nets["f"]["layers"][-1][2], m._prior_theta.cpu().numpy().ravel(), adrf.ravel(),
                       interval.ravel()]).astype(np.float32)
t = torch.from_numpy(flat).cuda()
mx, mn = t.clone(), t.clone()
dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
out = dict(rank=dist.get_rank(), spread=float((mx - mn).abs().max().item()), finite=bool(np.all(np.isfinite(flat))), rows=int(m.data_z.shape[0]),
           adrf=[float(a) for a in adrf], adrf_untrained=[float(a) for a in adrf_u], interval_untrained=[float(a) for a in interval_u.ravel()],
           adrf_untrained_adaptive=[float(a) for a in adrf_a], q_sd_adapted=q_sd_a, predict_sharded=sharded, loss=[h["loss_postrior_z"] for h in m.fit_history], kl_prior=[h["kl_prior"] for h in m.fit_history])
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dp_print import print_in_rank_order
print_in_rank_order(json.dumps(out))
assert out["spread"] == 0.0 and out["finite"]
dist.destroy_process_group()
