"""The RCCL communicator behind the C ABI (csrc/comm_api.hip) and the data-parallel epoch calls that enqueue their gradient
all-reduce on it (bgm_causal_fit_epoch_dp, bgm_bnn_fit_epoch_dp) -- on ONE device, with a one-rank communicator: RCCL is loaded,
a communicator is created, ncclAllReduce runs on the library's stream between the gradient kernels and the Adam step, and the
results equal the single-process epoch call / the per-minibatch host loop bit for bit.  (Two ranks on two devices:
tests/test_gpu_rccl.py.)  reference: the loop that is sharded is causalbgm/base.py:488-514; the reference has no collective."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "hirano_imbens_N2000_p20_seed0.npz")


def _flat(net):
    return np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for W, b in net])


def _comm():
    from bayesgm_amd import parallel
    return parallel.DeviceComm(torch.device("cuda", 0), world=1, rank_=0)


def test_one_rank_communicator_and_all_reduce_on_the_stream():
    """bgm_comm_unique_id / _create / _info / _all_reduce_f32 / _destroy: the library resolves RCCL at run time (the copy torch maps),
    a one-rank sum leaves the buffer as it is, and the call is ordered with the kernels around it on the stream."""
    c = _comm()
    info = c.info()
    assert info["world"] == 1 and info["rank"] == 0 and "rccl" in info["library"]
    print("RCCL:", info["library"])
    t = torch.arange(40000, device="cuda", dtype=torch.float32)
    t.mul_(2.0)                      # a kernel in front of the collective on the same stream
    c.all_reduce_sum_(t)
    t.add_(1.0)                      # ... and one behind it
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(40000, dtype=torch.float32) * 2 + 1)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):       # a side stream: the collective follows torch's CURRENT stream
        u = torch.full((1000,), 3.0, device="cuda")
        c.all_reduce_sum_(u)
        u.mul_(2.0)
    s.synchronize()
    assert float(u.sum().item()) == 6000.0
    c.close()


def test_comm_argument_errors():
    import ctypes as C
    from bayesgm_amd import _lib
    lib = _lib.load()
    ident = np.zeros(128, np.uint8)
    h = C.c_void_p()
    assert lib.bgm_comm_create(0, ident.ctypes.data_as(C.c_void_p), 2, 2, C.byref(h)) != 0          # rank outside the world
    assert lib.bgm_comm_create(0, None, 1, 0, C.byref(h)) != 0
    assert lib.bgm_comm_all_reduce_f32(None, None, 4, None) != 0
    assert lib.bgm_comm_destroy(None) == 0


@pytest.mark.parametrize("z_adam", ["replay", "lazy", "dense"])
def test_data_parallel_epoch_call_with_one_rank_equals_the_single_process_loops(z_adam, tmp_path):
    """CausalBGM.fit(dp_comm=one-rank communicator): bgm_causal_fit_epoch_dp -- gradient tiles, ncclAllReduce, a separate Adam launch,
    the latent phase on the second stream under HIP events -- gives the networks and the latent table of the fused single-process
    epoch call and of the per-minibatch host loop bit for bit (2000 = 62 x 32 + 16: the short last minibatch included)."""
    from bayesgm_amd.models import CausalBGM
    g = np.load(GOLD)
    x, y, v = g["x"], g["y"], g["v"]
    params = dict(dataset="t", output_dir=str(tmp_path), save_res=False, save_model=False, binary_treatment=False, use_bnn=False,
                  z_dims=[1, 1, 1, 7], v_dim=v.shape[1], lr_theta=1e-3, lr_z=1e-3, lr=2e-4, g_d_freq=5, use_z_rec=True, kl_weight=1e-4,
                  g_units=[64] * 5, e_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8], dz_units=[64, 32, 8])
    comm = _comm()
    res = []
    for kw in (dict(host_loop=True), dict(host_loop=False), dict(dp_comm=comm)):
        model = CausalBGM(dict(params), random_seed=5)
        model.fit((x, y, v), epochs=2, epochs_per_eval=1, batch_size=32, use_egm_init=False, verbose=0, z_adam=z_adam, **kw)
        res.append((model.data_z.cpu().numpy().copy(), {k: _flat(model.nets[k]) for k in "gfh"}, [dict(h) for h in model.fit_history]))
    comm.close()
    for other in res[:2]:
        assert np.array_equal(other[0], res[2][0])
        for k in "gfh":
            assert np.array_equal(other[1][k], res[2][1][k]), k
    for a, b in zip(res[0][2], res[2][2]):
        for key in ("loss_v", "loss_x", "loss_y", "loss_postrior_z", "mse_v", "mse_y"):
            assert abs(a[key] - b[key]) <= 1e-6 * max(1.0, abs(a[key])), (key, a[key], b[key])


def test_data_parallel_epoch_call_outside_the_row_tile_chains(tmp_path):
    """minibatches of 100 rows (the general forward / backward kernels, no second stream): same identity"""
    from bayesgm_amd.models import CausalBGM
    g = np.load(GOLD)
    x, y, v = g["x"][:700], g["y"][:700], g["v"][:700]
    params = dict(dataset="t", output_dir=str(tmp_path), save_res=False, save_model=False, binary_treatment=False, use_bnn=False,
                  z_dims=[1, 1, 1, 7], v_dim=v.shape[1], lr_theta=1e-3, lr_z=1e-3, lr=2e-4, g_d_freq=5, use_z_rec=True, kl_weight=1e-4,
                  g_units=[64] * 5, e_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8], dz_units=[64, 32, 8])
    comm = _comm()
    res = []
    for kw in (dict(host_loop=True), dict(dp_comm=comm)):
        model = CausalBGM(dict(params), random_seed=6)
        model.fit((x, y, v), epochs=1, epochs_per_eval=1, batch_size=100, use_egm_init=False, verbose=0, **kw)
        res.append((model.data_z.cpu().numpy().copy(), {k: _flat(model.nets[k]) for k in "gfh"}))
    comm.close()
    assert np.array_equal(res[0][0], res[1][0])
    for k in "gfh":
        assert np.array_equal(res[0][1][k], res[1][1][k]), k


@pytest.mark.parametrize("z_adam", ["replay", "dense"])
def test_bayesian_data_parallel_epoch_call_with_one_rank(z_adam, tmp_path):
    """CausalBGM(use_bnn=True).fit(dp_comm=...): bgm_bnn_fit_epoch_dp (theta gradient with apply = 0, ncclAllReduce of the session's
    gradient, bgm_bnn_theta_apply, latent step; from C++) against the per-minibatch host loop: parameters, latents and the noise-stream
    counter are equal bit for bit (n = 200 = 6 x 32 + 8)."""
    from bayesgm_amd.models import CausalBGM
    rs = np.random.RandomState(0)
    n, p = 200, 100
    v = rs.randn(n, p).astype(np.float32)
    x = rs.exponential(size=(n, 1)).astype(np.float32)
    y = (x + 0.3 * v[:, :1] + rs.randn(n, 1)).astype(np.float32)
    params = dict(dataset="t", output_dir=str(tmp_path), save_res=False, save_model=False, binary_treatment=False, use_bnn=True,
                  z_dims=[1, 1, 1, 7], v_dim=p, lr_theta=1e-3, lr_z=1e-3, lr=2e-4, g_d_freq=5, use_z_rec=True, kl_weight=1e-4,
                  g_units=[64] * 5, e_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8], dz_units=[64, 32, 8])
    comm = _comm()
    res = []
    for i, kw in enumerate((dict(host_loop=True), dict(dp_comm=comm))):
        m = CausalBGM(dict(params), timestamp="t%d" % i, random_seed=3)
        m.fit((x, y, v), epochs=2, epochs_per_eval=1, batch_size=32, use_egm_init=False, verbose=0, z_adam=z_adam, **kw)
        res.append((m.data_z.cpu().numpy().copy(), m.engine.read(0).copy(), m._stream))
    comm.close()
    (za, ta, sa), (zb, tb, sb) = res
    assert sa == sb
    # the host loop of ONE process applies Adam inside the gradient kernel (apply = 1), the data-parallel call as its own launch
    # (apply = 0 + bgm_bnn_theta_apply): one expression (bnn_adam_one), so the parameters agree to the last bit
    assert np.array_equal(ta, tb) and np.array_equal(za, zb)
