"""YAML-config driver (bayesgm_amd/main.py) with config files that carry the reference's keys."""
import numpy as np
import pytest

from bayesgm_amd import main as M

CAUSAL = """
dataset: Sim_Hirano_Imbens
output_dir: '%s'
save_res: False
save_model: False
binary_treatment: False
use_bnn: %s
z_dims: [1,1,1,3]
v_dim: 12
lr_theta: 0.0001
lr_z: 0.0001
g_units: [64,64,64,64,64]
f_units: [64,32,8]
h_units: [64,32,8]
kl_weight: 0.0001
lr: 0.0002
g_d_freq: 2
use_z_rec: True
e_units: [64,64,64,64,64]
dz_units: [64,32,8]
"""

BGM = """
dataset: Sim_heteroskedastic
output_dir: '%s'
save_res: False
save_model: False
use_bnn: %s
rank: 2
z_dim: 3
x_dim: 8
lr_theta: 0.005
lr_z: 0.005
g_units: [64,64,64]
kl_weight: 0.00005
lr: 0.001
g_d_freq: 1
use_z_rec: True
alpha: 0.0
gamma: 0
e_units: [64,64,64]
dz_units: [64,32,8]
dx_units: [64,32,8]
"""


def test_config_is_loaded_unchanged(tmp_path):
    f = tmp_path / "c.yaml"
    f.write_text(CAUSAL % (tmp_path, "True"))
    p = M.load_config(str(f))
    assert p["z_dims"] == [1, 1, 1, 3] and p["use_bnn"] is True and p["lr"] == 0.0002 and p["dataset"] == "Sim_Hirano_Imbens"
    a = M.build_parser().parse_args(["-c", str(f)])
    assert (a.n_rows, a.epochs, a.batches, a.burn_in) == (20000, None, None, 5000)
    g = tmp_path / "other.yaml"
    g.write_text("dataset: Mnist\noutput_dir: '.'\n")
    with pytest.raises(ValueError):
        M.main(["-c", str(g)])


@pytest.mark.gpu
@pytest.mark.parametrize("bnn", ["True", "False"])
def test_causal_workflow_from_yaml(tmp_path, bnn):
    f = tmp_path / "c.yaml"
    f.write_text(CAUSAL % (tmp_path, bnn))
    model, adrf, itv = M.main(["-c", str(f), "-n", "400", "-e", "2", "-b", "8", "--epochs_per_eval", "1", "--egm_batches_per_eval", "4",
                               "--n_mcmc", "20", "--burn_in", "20", "--seed", "1"])
    assert adrf.shape == (20,) and itv.shape == (20, 2) and np.isfinite(adrf).all()
    assert type(model).__name__ == ("CausalBGMBayes" if bnn == "True" else "CausalBGM")


@pytest.mark.gpu
@pytest.mark.parametrize("bnn", ["True", "False"])
def test_bgm_workflow_from_yaml(tmp_path, bnn):
    f = tmp_path / "b.yaml"
    f.write_text(BGM % (tmp_path, bnn))
    model, imp, itv = M.main(["-c", str(f), "-n", "600", "-e", "2", "-b", "8", "--epochs_per_eval", "1", "--egm_batches_per_eval", "4",
                              "--n_mcmc", "20", "--burn_in", "20", "--seed", "1"])
    assert imp.shape == (60, 8) and np.isfinite(imp).all() and np.asarray(itv).shape == (60, 1, 2)
    assert type(model).__name__ == ("BGMBayes" if bnn == "True" else "BGM")
