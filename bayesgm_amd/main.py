"""YAML-config driver: ``python -m bayesgm_amd.main -c CONFIG.yaml [...]``.

Counterpart of /root/reference/src/main.py:18-87 (the reference's research driver): a config file with the keys of
src/configs/*.yaml (``dataset``, ``output_dir``, ``use_bnn``, ``z_dims`` / ``z_dim``, unit lists, learning rates ...) is
loaded with ``yaml.safe_load`` and handed to the model class unchanged; the two simulated workflows of the reference are
wired: ``Sim_Hirano_Imbens`` (CausalBGM: fit, ADRF on linspace(0, 3, 20)) and ``Sim_heteroskedastic`` (BGM: fit on 90 % of
the rows, impute the response of the held-out rows, main.py:66-84).  The reference hard-codes the run lengths; here they are
flags whose defaults are the reference's values.  No config files are shipped: the reference's own YAML files (or any file with
their keys, plus the build options `bnn_norm` / `bnn_mcmc_noise`) are read as they are; tests/test_main_yaml.py writes small ones.
"""
import argparse

import numpy as np
import yaml


def load_config(path):
    with open(path, "r") as f:
        return yaml.safe_load(f)


def build_parser():
    ap = argparse.ArgumentParser("bayesgm_amd.main")
    ap.add_argument("-c", "--config", type=str, required=True, help="path to the YAML config")
    ap.add_argument("-n", "--n_rows", type=int, default=20000, help="rows of the simulated panel (main.py:49,66)")
    ap.add_argument("-e", "--epochs", type=int, default=None, help="epochs (default: 100 causal / 200 BGM)")
    ap.add_argument("-b", "--batches", type=int, default=None, help="EGM iterations (default: 30000 causal / 50000 BGM)")
    ap.add_argument("--epochs_per_eval", type=int, default=10)
    ap.add_argument("--egm_batches_per_eval", type=int, default=500)
    ap.add_argument("--n_mcmc", type=int, default=None, help="retained draws (default: 3000 causal / 5000 BGM)")
    ap.add_argument("--burn_in", type=int, default=5000)
    ap.add_argument("--seed", type=int, default=None)
    return ap


def run(args):
    params = load_config(args.config)
    ds = params["dataset"]
    if ds == "Sim_Hirano_Imbens":
        from .datasets import Sim_Hirano_Imbens_sampler
        from .models import CausalBGM
        from .utils import get_ADRF
        x, y, v = Sim_Hirano_Imbens_sampler(N=args.n_rows, v_dim=params["v_dim"]).load_all()
        model = CausalBGM(params=params, random_seed=args.seed)
        model.fit(data=(x, y, v), epochs=args.epochs or 100, epochs_per_eval=args.epochs_per_eval, use_egm_init=True,
                  egm_n_iter=args.batches if args.batches is not None else 30000, egm_batches_per_eval=args.egm_batches_per_eval, verbose=1)
        xs = np.linspace(0, 3, 20)
        causal_pre, pos_intervals = model.predict(data=(x, y, v), alpha=0.01, n_mcmc=args.n_mcmc or 3000, burn_in=args.burn_in,
                                                  x_values=xs, q_sd=1.0)
        truth = get_ADRF(x_values=list(xs), dataset="Imbens")
        print("ADRF RMSE vs truth: %.4f" % float(np.sqrt(np.mean((causal_pre - truth) ** 2))))
        return model, causal_pre, pos_intervals
    if ds == "Sim_heteroskedastic":
        from sklearn.model_selection import train_test_split
        from .datasets import simulate_z_hetero
        from .models import BGM
        X, Y = simulate_z_hetero(n=args.n_rows, k=params["z_dim"], d=params["x_dim"] - 1)
        X_train, X_test, Y_train, Y_test = train_test_split(X, Y, test_size=0.1, random_state=123)
        data_train = np.c_[X_train, Y_train].astype("float32")
        model = BGM(params=params, random_seed=args.seed)
        model.fit(data=data_train, epochs=args.epochs or 200, epochs_per_eval=args.epochs_per_eval, use_egm_init=True,
                  egm_n_iter=args.batches if args.batches is not None else 50000, egm_batches_per_eval=args.egm_batches_per_eval, verbose=1)
        data_test = np.hstack([X_test.astype("float32"), np.full((X_test.shape[0], 1), np.nan, dtype=np.float32)])
        data_x_pred, pred_interval = model.predict(data=data_test, alpha=0.05, bs=100, n_mcmc=args.n_mcmc or 5000, burn_in=args.burn_in,
                                                   step_size=0.01, num_leapfrog_steps=10, seed=42)
        pcc = float(np.corrcoef(np.asarray(Y_test, np.float64).ravel(), data_x_pred[:, -1].astype(np.float64))[0, 1])
        print("held-out response: Pearson correlation of the posterior-mean imputation %.4f" % pcc)
        return model, data_x_pred, pred_interval
    raise ValueError("bayesgm_amd.main: dataset %r has no simulated workflow here (Sim_Hirano_Imbens, Sim_heteroskedastic); "
                     "use `python -m bayesgm_amd.cli` for data files" % ds)


def main(argv=None):
    return run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
