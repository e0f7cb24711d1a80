"""Generate golden fixtures by IMPORTING the reference (run in the build container only).

    python tests/golden/make_golden.py

Writes small .npz files holding inputs/outputs of the reference's own
``bayesgm.datasets`` / ``bayesgm.utils`` (the only reference modules importable
without TensorFlow).  The fixtures are data; the reference source never enters
this repository and never travels to the GPU box.
"""
import hashlib
import os
import sys

import numpy as np

REF = "/root/reference/src"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    from bayesgm.datasets import Sim_Hirano_Imbens_sampler, Gaussian_sampler, simulate_z_hetero, Base_sampler
    from bayesgm.utils.helpers import get_ADRF

    # C0-sized Hirano-Imbens panel (N=2000, p=20) -- full arrays
    s = Sim_Hirano_Imbens_sampler(N=2000, v_dim=20, seed=0)
    x, y, v = s.load_all()
    b1 = s.next_batch()
    b2 = s.next_batch()
    np.savez_compressed(os.path.join(HERE, "hirano_imbens_N2000_p20_seed0.npz"),
                        x=x, y=y, v=v, batch1_x=b1[0], batch2_v=b2[2], full_index=s.full_index)
    # larger panels -- hashes + head only (arrays too large to commit)
    big = {}
    for (N, p, seed) in [(20000, 200, 0), (5000, 100, 3)]:
        s = Sim_Hirano_Imbens_sampler(N=N, v_dim=p, seed=seed)
        x, y, v = s.load_all()
        big[f"N{N}_p{p}_s{seed}_sha_x"] = sha(x)
        big[f"N{N}_p{p}_s{seed}_sha_y"] = sha(y)
        big[f"N{N}_p{p}_s{seed}_sha_v"] = sha(v)
        big[f"N{N}_p{p}_s{seed}_head_v"] = v[:4, :8]
        big[f"N{N}_p{p}_s{seed}_head_x"] = x[:8, 0]
        big[f"N{N}_p{p}_s{seed}_head_y"] = y[:8, 0]
    np.savez_compressed(os.path.join(HERE, "hirano_imbens_hashes.npz"), **big)

    xs = np.linspace(0, 3, 20)
    adrf = {f"adrf_{d}": get_ADRF(x_values=list(xs), dataset=d) for d in ("Imbens", "Sun", "Lee")}
    adrf["x_values"] = xs
    adrf["adrf_range_Imbens"] = get_ADRF(x_min=0.5, x_max=2.5, nb_intervals=7, dataset="Imbens")
    np.savez_compressed(os.path.join(HERE, "adrf_truth.npz"), **adrf)

    # ---- data_io: parse_file / parse_file_triplet on tiny input files kept beside the fixtures (utils/data_io.py:33-150)
    from bayesgm.utils.data_io import parse_file, parse_file_triplet, save_data
    io_dir = os.path.join(HERE, "io")
    os.makedirs(io_dir, exist_ok=True)
    rs = np.random.RandomState(11)
    mat = np.round(rs.randn(12, 5) * np.array([1.0, 10.0, 0.1, 3.0, 1.0]) + np.array([0, 5, -2, 0, 1.0]), 4)
    mat[:, 4] = 2.5                                        # a constant column (StandardScaler edge case)
    np.savetxt(os.path.join(io_dir, "mat_tab.txt"), mat, fmt="%.4f", delimiter="\t")
    with open(os.path.join(io_dir, "mat_comma.csv"), "w") as f:
        f.write("x,y,v1,v2,v3\n")
        for row in mat:
            f.write(",".join("%.4f" % t for t in row) + "\n")
    np.savez(os.path.join(io_dir, "mat_keys.npz"), other=mat[:3], x=mat)       # 'x' is preferred over the first key
    np.savez(os.path.join(io_dir, "mat_first.npz"), foo=mat[:7], bar=mat)      # no known key: first key wins
    np.savez(os.path.join(io_dir, "triplet.npz"), x=mat[:, :1], y=mat[:, 1:2], v=mat[:, 2:])
    io = {}
    for name, kw in [("mat_tab.txt", dict(sep="\t")), ("mat_comma.csv", dict(sep=",")), ("mat_keys.npz", {}), ("mat_first.npz", {})]:
        for norm in (True, False):
            io[f"parse_{name}_{int(norm)}"] = parse_file(os.path.join(io_dir, name), normalize=norm, **kw)
    for name, kw in [("mat_tab.txt", dict(sep="\t")), ("mat_comma.csv", dict(sep=",")), ("triplet.npz", {})]:
        for norm in (True, False):
            x_, y_, v_ = parse_file_triplet(os.path.join(io_dir, name), normalize=norm, **kw)
            io[f"triplet_{name}_{int(norm)}_x"], io[f"triplet_{name}_{int(norm)}_y"], io[f"triplet_{name}_{int(norm)}_v"] = x_, y_, v_
    save_data(os.path.join(io_dir, "saved.txt"), mat[:4])
    save_data(os.path.join(io_dir, "saved.csv"), mat[:4], delimiter=",")
    np.savez_compressed(os.path.join(HERE, "data_io.npz"), **io)

    gs = Gaussian_sampler(mean=np.zeros(10), sd=1.0)
    np.random.seed(5)
    gb = gs.get_batch(32)
    np.savez_compressed(os.path.join(HERE, "gaussian_sampler.npz"), X_head=gs.X[:16], batch=gb,
                        X_sha=np.array(sha(gs.X)))

    X, Y = simulate_z_hetero(n=2000, k=3, d=19, seed=42)
    np.savez_compressed(os.path.join(HERE, "z_hetero_n2000.npz"), X=X.astype(np.float32),
                        Y=Y.astype(np.float32), X_sha64=np.array(sha(X)), Y_sha64=np.array(sha(Y)))

    # the other continuous-treatment simulators of the YAML configs (Sim_Sun.yaml, Sim_Colangelo.yaml)
    from bayesgm.datasets import Sim_Sun_sampler, Sim_Colangelo_sampler
    sims = {}
    for name, cls, kw in (("sun", Sim_Sun_sampler, dict(N=1500, v_dim=12, seed=2)),
                          ("colangelo", Sim_Colangelo_sampler, dict(N=1200, v_dim=9, seed=4))):
        sx, sy, sv = cls(**kw).load_all()
        sims[name + "_x"], sims[name + "_y"], sims[name + "_v"] = sx, sy, sv
    np.savez_compressed(os.path.join(HERE, "sun_colangelo.npz"), **sims)

    # Base_sampler batching on a tiny deterministic panel, incl. the wrap-around batch
    xx = np.arange(10, dtype=np.float32)
    bs = Base_sampler(xx, xx * 2, np.stack([xx, -xx], 1), batch_size=4, normalize=False)
    batches = np.stack([bs.next_batch()[0][:, 0] for _ in range(7)])
    np.savez_compressed(os.path.join(HERE, "base_sampler_batches.npz"), batches=batches)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
