"""Command line of the build: ``python -m bayesgm_amd.cli causalbgm|bgm -i DATA -o OUT [...]``.

Same sub-commands, flags, defaults, workflow (fit -> predict -> result files) and output file names as the reference's
``bayesgm`` console script (/root/reference/src/bayesgm/cli/cli.py:8-262), so a command line
written for the reference runs unchanged.  The option tables below are the flag set of that file; parsing is argparse.
"""
import argparse

import numpy as np

VERSION = "1.0.2-mi355x"
_BOOL = argparse.BooleanOptionalAction
_UNITS5, _UNITS3 = [64] * 5, [64, 32, 8]

# (flags, keyword arguments) -- shared by both sub-commands (cli.py:8-28)
_COMMON = [
    (("-o", "--output_dir"), dict(type=str, required=True, help="Output directory")),
    (("-i", "--input"), dict(type=str, required=True, help="Input data file (csv, txt or npz)")),
    (("-t", "--delimiter"), dict(type=str, default="\t", help="Delimiter of txt / csv files (default: tab)")),
    (("-d", "--dataset"), dict(type=str, default="Mydata", help="Dataset name")),
    (("-F", "--save_format"), dict(type=str, default="txt", help="Format of the saved results (default: txt)")),
    (("-save_model",), dict(default=False, action=_BOOL, help="Save the model")),
    (("-save_res",), dict(default=True, action=_BOOL, help="Save intermediate results")),
    (("--use_bnn",), dict(default=True, action=_BOOL, help="Bayesian neural nets")),
    (("--use_egm_init",), dict(default=True, action=_BOOL, help="EGM initialisation")),
    (("--seed",), dict(type=int, default=123, help="Random seed (default: 123)")),
]

# CausalBGM (cli.py:31-94)
_CAUSAL = [
    (("-B", "--binary_treatment"), dict(default=True, action=_BOOL, help="Binary treatment")),
    (("-Z", "--z_dims"), dict(type=int, nargs="+", default=[3, 3, 6, 6], help="Latent dimensions (default: 3 3 6 6)")),
    (("--lr_theta",), dict(type=float, default=1e-4, help="Learning rate of the network parameters")),
    (("--lr_z",), dict(type=float, default=1e-4, help="Learning rate of the latent variables")),
    (("--x_values",), dict(type=float, nargs="+", help="Treatment values to predict at (continuous treatment)")),
    (("--g_units",), dict(type=int, nargs="+", default=_UNITS5, help="Hidden units of the covariate model")),
    (("--f_units",), dict(type=int, nargs="+", default=_UNITS3, help="Hidden units of the outcome model")),
    (("--h_units",), dict(type=int, nargs="+", default=_UNITS3, help="Hidden units of the treatment model")),
    (("--kl_weight",), dict(type=float, default=1e-4, help="Weight of the KL term of the Bayesian nets")),
    (("--lr",), dict(type=float, default=1e-4, help="Learning rate of the EGM initialisation")),
    (("--g_d_freq",), dict(type=int, default=5, help="Discriminator steps per generator step")),
    (("--e_units",), dict(type=int, nargs="+", default=_UNITS5, help="Hidden units of the encoder")),
    (("--dz_units",), dict(type=int, nargs="+", default=_UNITS3, help="Hidden units of the latent discriminator")),
    (("--use-z-rec",), dict(default=True, action=_BOOL, help="Latent reconstruction term")),
    (("-N", "--n_iter"), dict(type=int, default=30000, help="EGM iterations (default: 30000)")),
    (("--startoff",), dict(type=int, default=0, help="First epoch considered for the best model")),
    (("--batches_per_eval",), dict(type=int, default=500, help="EGM iterations per evaluation")),
    (("-E", "--epochs"), dict(type=int, default=100, help="Epochs of the iterative updates")),
    (("-M", "--n_mcmc"), dict(type=int, default=3000, help="Retained MCMC samples")),
    (("--burn_in",), dict(type=int, default=5000, help="Burn-in iterations of Metropolis-Hastings")),
    (("-q", "--q_sd"), dict(type=float, default=1.0, help="Proposal std-dev; negative = adaptive")),
    (("--epochs_per_eval",), dict(type=int, default=10, help="Epochs per evaluation")),
    (("--alpha",), dict(type=float, default=0.01, help="Significance level")),
]

# BGM (cli.py:97-162)
_BGM = [
    (("--z_dim",), dict(type=int, default=10, help="Latent dimension")),
    (("--g_units",), dict(type=int, nargs="+", default=_UNITS5, help="Hidden units of the generative model")),
    (("--e_units",), dict(type=int, nargs="+", default=_UNITS5, help="Hidden units of the encoder")),
    (("--dz_units",), dict(type=int, nargs="+", default=_UNITS3, help="Hidden units of the latent discriminator")),
    (("--dx_units",), dict(type=int, nargs="+", default=_UNITS3, help="Hidden units of the data discriminator")),
    (("--lr_theta",), dict(type=float, default=1e-4, help="Learning rate of the network parameters")),
    (("--lr_z",), dict(type=float, default=1e-4, help="Learning rate of the latent variables")),
    (("--lr",), dict(type=float, default=1e-4, help="Learning rate of the EGM initialisation")),
    (("--kl_weight",), dict(type=float, default=1e-4, help="Weight of the KL term of the Bayesian nets")),
    (("--g_d_freq",), dict(type=int, default=5, help="Discriminator steps per generator step")),
    (("--gamma",), dict(type=float, default=10.0, help="Gradient-penalty coefficient of the EGM discriminators")),
    (("--egm_reg_alpha",), dict(type=float, default=0.01, help="Variance regularisation of the EGM generator step")),
    (("-N", "--egm_n_iter"), dict(type=int, default=20000, help="EGM iterations (default: 20000)")),
    (("--egm_batches_per_eval",), dict(type=int, default=500, help="EGM iterations per evaluation")),
    (("-E", "--epochs"), dict(type=int, default=100, help="Epochs of the iterative updates")),
    (("--epochs_per_eval",), dict(type=int, default=5, help="Epochs per evaluation")),
    (("--batch_size",), dict(type=int, default=32, help="Minibatch size")),
    (("--alpha",), dict(type=float, default=0.05, help="Significance level of the prediction intervals")),
    (("-M", "--n_mcmc"), dict(type=int, default=5000, help="Retained MCMC samples")),
    (("--burn_in",), dict(type=int, default=5000, help="Burn-in iterations")),
    (("--step_size",), dict(type=float, default=0.01, help="HMC step size")),
    (("--num_leapfrog_steps",), dict(type=int, default=10, help="Leapfrog steps per HMC transition")),
]


def _fill(parser, *tables):
    for table in tables:
        for flags, kw in table:
            parser.add_argument(*flags, **kw)


def run_causalbgm(args):
    """cli.py:165-210: parse the (x, y, v) triplet, fit, predict, save point estimate and interval."""
    from .models import CausalBGM
    from .utils import parse_file_triplet, save_data
    params = {k: v for k, v in vars(args).items() if k not in ("func", "command")}
    data = parse_file_triplet(args.input, sep=params["delimiter"])
    params["v_dim"] = data[-1].shape[1]
    model = CausalBGM(params=params, random_seed=None)
    model.fit(data=data, epochs=params["epochs"], epochs_per_eval=params["epochs_per_eval"], startoff=params["startoff"],
              use_egm_init=params["use_egm_init"], egm_n_iter=params["n_iter"], egm_batches_per_eval=params["batches_per_eval"],
              verbose=1)
    kw = dict(data=data, alpha=params["alpha"], n_mcmc=params["n_mcmc"], burn_in=params["burn_in"], q_sd=params["q_sd"])
    if not params["binary_treatment"]:
        kw["x_values"] = params["x_values"]
    causal_pre, pos_intervals = model.predict(**kw)
    save_data("{}/causal_effect_point_estimate.{}".format(model.save_dir, params["save_format"]), causal_pre)
    save_data("{}/causal_effect_posterior_interval.{}".format(model.save_dir, params["save_format"]), pos_intervals)
    return model


def run_bgm(args):
    """cli.py:213-255: parse the data matrix, fit, impute with intervals, save."""
    from .models import BGM
    from .utils import parse_file, save_data
    params = {k: v for k, v in vars(args).items() if k not in ("func", "command")}
    data = parse_file(args.input, sep=params["delimiter"])
    params["x_dim"] = data.shape[1]
    predict_alpha = params.pop("alpha")              # significance level of predict ...
    params["alpha"] = params.pop("egm_reg_alpha")    # ... vs the model's EGM variance regularisation 'alpha'
    model = BGM(params=params, random_seed=params.get("seed"))
    model.fit(data=data, batch_size=params["batch_size"], epochs=params["epochs"], epochs_per_eval=params["epochs_per_eval"],
              use_egm_init=params["use_egm_init"], egm_n_iter=params["egm_n_iter"],
              egm_batches_per_eval=params["egm_batches_per_eval"], verbose=1)
    data_imputed, pred_interval = model.predict(data=data, alpha=predict_alpha, n_mcmc=params["n_mcmc"], burn_in=params["burn_in"],
                                                step_size=params["step_size"], num_leapfrog_steps=params["num_leapfrog_steps"],
                                                seed=params.get("seed", 42))
    save_data("{}/imputed_data.{}".format(model.save_dir, params["save_format"]), data_imputed)
    np.savez("{}/prediction_intervals.npz".format(model.save_dir), intervals=pred_interval)
    return model


def build_parser():
    parser = argparse.ArgumentParser("bayesgm", description="BayesGM on MI355X (bayesgm_amd) - v%s" % VERSION)
    parser.add_argument("--version", action="version", version="%(prog)s " + VERSION)
    sub = parser.add_subparsers(title="commands", description="Available model commands", dest="command")
    pc = sub.add_parser("causalbgm", help="CausalBGM: causal inference in observational studies")
    _fill(pc, _COMMON, _CAUSAL)
    pc.set_defaults(func=run_causalbgm)
    pb = sub.add_parser("bgm", help="BGM: data generation and missing-data imputation")
    _fill(pb, _COMMON, _BGM)
    pb.set_defaults(func=run_bgm)
    return parser


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.command is None:
        parser.print_help()
        return None
    return args.func(args)


def main_causalbgm(argv=None):
    """The flat ``causalBGM`` entry point kept by the reference for backwards compatibility (cli.py:288-344)."""
    parser = argparse.ArgumentParser("causalBGM", description="CausalBGM on MI355X (bayesgm_amd) - v%s" % VERSION)
    _fill(parser, _COMMON, _CAUSAL)
    args = parser.parse_args(argv)
    return run_causalbgm(args)


if __name__ == "__main__":
    main()
