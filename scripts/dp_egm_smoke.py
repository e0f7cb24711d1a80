"""Data-parallel EGM warm start, two ranks (deterministic and Bayesian networks).  On a 1-GPU box:
   BGM_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29536 \\
       scripts/dp_egm_smoke.py gloo [out.npz]
Every rank holds only its rows; the dz and the fused g | e | f | h gradients are all-reduced each step, so the ranks must end with
identical networks.  Rank 0 writes them to out.npz; tests/test_gpu_egm.py compares with ONE process stepping on the same global
minibatches (CausalBGM._egm_emulate_world)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.distributed as dist
backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
out_path = sys.argv[2] if len(sys.argv) > 2 else None
dev = int(os.environ.get("BGM_DEVICE", os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
dist.init_process_group(backend=backend)
from dp_egm_common import build, flat_weights, DATA, N_ITER, BATCH, PER_EVAL
res, saved = {}, {}
for use_bnn in (False, True):
    m = build(use_bnn, dev)
    w0 = flat_weights(m)
    m.egm_init(DATA, egm_n_iter=N_ITER, batch_size=BATCH, egm_batches_per_eval=PER_EVAL, verbose=0)
    w = flat_weights(m)
    t = torch.from_numpy(w).cuda()
    mx, mn = t.clone(), t.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
    key = "bnn" if use_bnn else "det"
    res[key] = dict(spread=float((mx - mn).abs().max().item()), moved=float(np.abs(w - w0).max()), finite=bool(np.all(np.isfinite(w))),
                    late_l2z=m._egm_late_l2z)
    saved[key] = w
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dp_print import print_in_rank_order
print_in_rank_order(json.dumps(dict(rank=dist.get_rank(), **res)))
if out_path and dist.get_rank() == 0:
    np.savez(out_path, **saved)
assert all(r["spread"] == 0.0 and r["finite"] for r in res.values())
dist.destroy_process_group()
