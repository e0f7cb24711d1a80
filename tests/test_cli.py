"""Command line: same flags and defaults as the reference's console script; the end-to-end run needs a GPU."""
import numpy as np
import pytest

from bayesgm_amd import cli


def test_causalbgm_defaults_are_the_reference_defaults():
    a = cli.build_parser().parse_args(["causalbgm", "-i", "in.csv", "-o", "out"])
    assert (a.use_bnn, a.use_egm_init, a.binary_treatment, a.save_res, a.save_model) == (True, True, True, True, False)
    assert a.z_dims == [3, 3, 6, 6] and a.g_units == [64] * 5 and a.f_units == [64, 32, 8] and a.dz_units == [64, 32, 8]
    assert (a.n_iter, a.epochs, a.n_mcmc, a.burn_in, a.q_sd, a.alpha, a.epochs_per_eval) == (30000, 100, 3000, 5000, 1.0, 0.01, 10)
    assert (a.lr, a.lr_theta, a.lr_z, a.kl_weight, a.g_d_freq, a.use_z_rec, a.seed, a.delimiter) == (1e-4, 1e-4, 1e-4, 1e-4, 5, True, 123, "\t")
    b = cli.build_parser().parse_args(["causalbgm", "-i", "x", "-o", "o", "--no-use_bnn", "--no-binary_treatment", "--x_values", "0.5", "1",
                                       "-Z", "1", "1", "1", "7", "-N", "10", "-E", "2", "-M", "5", "-q", "-1", "-t", ","])
    assert (b.use_bnn, b.binary_treatment, b.x_values, b.z_dims, b.n_iter, b.epochs, b.n_mcmc, b.q_sd, b.delimiter) == \
        (False, False, [0.5, 1.0], [1, 1, 1, 7], 10, 2, 5, -1.0, ",")


def test_bgm_defaults_are_the_reference_defaults():
    a = cli.build_parser().parse_args(["bgm", "-i", "in.csv", "-o", "out"])
    assert (a.z_dim, a.egm_n_iter, a.epochs, a.epochs_per_eval, a.batch_size, a.n_mcmc, a.burn_in) == (10, 20000, 100, 5, 32, 5000, 5000)
    assert (a.alpha, a.gamma, a.egm_reg_alpha, a.step_size, a.num_leapfrog_steps, a.dx_units) == (0.05, 10.0, 0.01, 0.01, 10, [64, 32, 8])


def test_no_command_prints_help(capsys):
    assert cli.main([]) is None
    assert "causalbgm" in capsys.readouterr().out


@pytest.mark.gpu
@pytest.mark.parametrize("bnn", [True, False])
def test_causalbgm_command_end_to_end(tmp_path, bnn):
    from bayesgm_amd.datasets import Sim_Hirano_Imbens_sampler
    x, y, v = Sim_Hirano_Imbens_sampler(N=400, v_dim=12, seed=0).load_all()
    f = tmp_path / "panel.csv"
    np.savetxt(f, np.hstack([x, y, v]), delimiter=",")
    argv = ["causalbgm", "-i", str(f), "-o", str(tmp_path), "-t", ",", "--no-binary_treatment", "-Z", "1", "1", "1", "3", "-N", "12",
            "--batches_per_eval", "6", "-E", "2", "--epochs_per_eval", "1", "-M", "20", "--burn_in", "20", "--x_values", "0.5", "1.5", "2.5",
            "--use_bnn" if bnn else "--no-use_bnn"]
    model = cli.main(argv)
    est = np.loadtxt("%s/causal_effect_point_estimate.txt" % model.save_dir)
    itv = np.loadtxt("%s/causal_effect_posterior_interval.txt" % model.save_dir)
    assert est.shape == (3,) and itv.shape == (3, 2) and np.isfinite(est).all() and np.all(itv[:, 0] <= itv[:, 1])
    assert type(model).__name__ == ("CausalBGMBayes" if bnn else "CausalBGM")


@pytest.mark.gpu
@pytest.mark.parametrize("bnn", [True, False])
def test_bgm_command_end_to_end(tmp_path, bnn):
    rs = np.random.RandomState(0)
    z = rs.standard_normal((320, 3))
    data = (z @ rs.standard_normal((3, 8)) + 0.1 * rs.standard_normal((320, 8))).astype(np.float32)
    f = tmp_path / "panel.csv"          # complete panel: the reference's command fits and imputes the same file (cli.py:213-255)
    np.savetxt(f, data, delimiter=",", header=",".join("c%d" % i for i in range(8)), comments="")
    argv = ["bgm", "-i", str(f), "-o", str(tmp_path), "-t", ",", "--z_dim", "3", "-N", "12", "--egm_batches_per_eval", "6", "-E", "2",
            "--epochs_per_eval", "1", "-M", "20", "--burn_in", "20", "--g_units", "64", "64", "64", "--e_units", "32", "32",
            "--use_bnn" if bnn else "--no-use_bnn"]
    model = cli.main(argv)
    imp = np.loadtxt("%s/imputed_data.txt" % model.save_dir)
    assert imp.shape == data.shape and np.isfinite(imp).all()
    from bayesgm_amd.utils import parse_file
    np.testing.assert_allclose(imp, parse_file(str(f), sep=","), rtol=1e-5, atol=1e-6)       # nothing missing: observed cells returned as given
    assert np.load("%s/prediction_intervals.npz" % model.save_dir)["intervals"].shape == (320, 0, 2)
    assert type(model).__name__ == ("BGMBayes" if bnn else "BGM")


def test_package_resolves_submodules_lazily():
    import bayesgm_amd as bayesgm
    assert bayesgm.models.CausalBGM.__name__ == "CausalBGM" and bayesgm.models.BGM.__name__ == "BGM"
    assert callable(bayesgm.datasets.Sim_Hirano_Imbens_sampler) and callable(bayesgm.utils.get_ADRF)
    with pytest.raises(AttributeError):
        bayesgm.no_such_module
