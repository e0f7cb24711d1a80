"""Python driver of the BGM session with the Bayesian generator (``use_bnn=True``; include/bgm_hip.h, bgm_bvn_*).

PyTorch supplies device memory and the HIP stream only; there is no CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .engine import _ptr, _f32

STREAM_PREDICT = 0x40000000      # + row block (oracle/bgm_bnn.py)
STREAM_DECODE = 0x50000000


def flatten_vnet(net):
    """{"gamma","beta","mean_mv","var_mv","trunk":[(loc,rho,bias)],"mean":(...),"var":(...)} -> flat float32, session order."""
    parts = [net["gamma"], net["beta"], net["mean_mv"], net["var_mv"]]
    for L in list(net["trunk"]) + [net["mean"], net["var"]]:
        parts += list(L)
    return np.concatenate([np.asarray(a, np.float32).ravel() for a in parts]).astype(np.float32)


def unflatten_vnet(theta, q, units, p):
    o = 0

    def take(shape):
        nonlocal o
        n = int(np.prod(shape))
        a = np.array(theta[o:o + n], dtype=np.float32).reshape(shape)
        o += n
        return a

    def layer(i, k):
        return (take((i, k)), take((i, k)), take((k,)))
    net = {"gamma": take((q,)), "beta": take((q,)), "mean_mv": take((q,)), "var_mv": take((q,))}
    dims = [q] + list(units)
    net["trunk"] = [layer(dims[i], dims[i + 1]) for i in range(len(dims) - 1)]
    net["mean"] = layer(dims[-1], p)
    net["var"] = layer(dims[-1], p)
    assert o == len(theta)
    return net


class BvnEngine(object):
    def __init__(self, x_dim, z_dim, g_units=(64,) * 5, kl_weight=5e-5, max_batch=32, device=0, hmc_frozen_noise=False):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("bayesgm_amd: no HIP device visible; the hot path has no CPU fallback")
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        torch.zeros(1, device=self.device)
        self.p, self.q, self.units = int(x_dim), int(z_dim), [int(u) for u in g_units]
        if len(self.units) + 2 > _lib.BGM_MAX_LAYERS:
            raise ValueError("g_units: at most %d hidden layers with use_bnn" % (_lib.BGM_MAX_LAYERS - 2))
        cfg = _lib.BvnConfig()
        cfg.x_dim, cfg.z_dim, cfg.n_hidden_g = self.p, self.q, len(self.units)
        for i, u in enumerate(self.units):
            cfg.g_units[i] = u
        cfg.kl_weight = float(kl_weight)
        cfg.max_batch = int(max_batch)
        cfg.hmc_frozen_noise = int(bool(hmc_frozen_noise))
        self.cfg = cfg
        n = C.c_int64()
        _lib.check(self.lib.bgm_bvn_layout(C.byref(cfg), C.byref(n)), "bgm_bvn_layout")
        self.n_params = int(n.value)
        self.h = C.c_void_p()
        _lib.check(self.lib.bgm_create(C.byref(self.h), self.device.index), "bgm_create")
        self.open = False

    def set_disc_norm(self, mode):
        """BatchNormalization mode of the EGM discriminators opened afterwards: "batch" | "fixed" (bgm_set_disc_norm)."""
        _lib.check(self.lib.bgm_set_disc_norm(self.h, {"batch": 0, "fixed": 1}[mode]), "bgm_set_disc_norm")

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.bgm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- session -------------------------------------------------------------
    def begin(self, net):
        theta = flatten_vnet(net) if isinstance(net, dict) else np.ascontiguousarray(net, np.float32)
        _lib.check(self.lib.bgm_bvn_begin(self.h, C.byref(self.cfg), theta.ctypes.data_as(C.c_void_p), theta.size, self._stream()),
                   "bgm_bvn_begin")
        self.open = True
        if getattr(self, "_precision", "fp32") != "fp32":      # a new session starts in fp32: the engine's mode travels with the engine
            self.set_precision(self._precision)

    def set_precision(self, mode="fp32"):
        """Arithmetic of hmc_run launched afterwards on this (open) session: "fp32" (default) | "f16x3" (split fp16 products with fp32
        accumulation on the streamed frozen-noise kernel, opt-in; bgm_bvn_set_precision).  "f16x3" raises for sessions it does not serve
        (fresh noise, z_dim > 16, hidden layers other than 3 or 5 of 64 units)."""
        _lib.check(self.lib.bgm_bvn_set_precision(self.h, {"fp32": 0, "f16x3": 2}[mode]), "bgm_bvn_set_precision")
        self._precision = mode

    def read(self, what=0):
        out = np.empty(self.n_params, np.float32)
        _lib.check(self.lib.bgm_bvn_read(self.h, int(what), out.ctypes.data_as(C.c_void_p), out.size, self._stream()), "bgm_bvn_read")
        return out

    def write(self, theta, what=0):
        theta = np.ascontiguousarray(theta, np.float32)
        _lib.check(self.lib.bgm_bvn_write(self.h, int(what), theta.ctypes.data_as(C.c_void_p), theta.size, self._stream()),
                   "bgm_bvn_write")

    def get_net(self):
        return unflatten_vnet(self.read(0), self.q, self.units, self.p)

    def end(self):
        if self.open:
            _lib.check(self.lib.bgm_bvn_end(self.h, self._stream()), "bgm_bvn_end")
            self.open = False

    # -- minibatch steps ---------------------------------------------------------
    def theta_step(self, x, data_z, idx, lr, seed, stream_id, batch_global=None, apply=True, out=None):
        B = int(idx.numel())
        _lib.check(self.lib.bgm_bvn_theta_step(self.h, _ptr(x), _ptr(data_z), _ptr(idx), B, int(batch_global or B), float(lr),
                                               int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream_id) & 0xFFFFFFFF, int(bool(apply)),
                                               _ptr(out), self._stream()), "bgm_bvn_theta_step")

    def grad_exchange(self, buf, to_session):
        _lib.check(self.lib.bgm_bvn_grad_exchange(self.h, _ptr(buf), int(bool(to_session)), self._stream()), "bgm_bvn_grad_exchange")

    def theta_apply(self, lr):
        _lib.check(self.lib.bgm_bvn_theta_apply(self.h, float(lr), self._stream()), "bgm_bvn_theta_apply")

    def z_step(self, x, data_z, idx, lr_z, seed, stream_id, batch_global=None, out=None):
        B = int(idx.numel())
        _lib.check(self.lib.bgm_bvn_z_step(self.h, _ptr(x), _ptr(data_z), _ptr(idx), B, int(batch_global or B), float(lr_z),
                                           int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream_id) & 0xFFFFFFFF, _ptr(out), self._stream()),
                   "bgm_bvn_z_step")

    # -- large batches (training=False) ----------------------------------------------
    def logpost(self, z, x, seed, stream_id, row_base=0, want_grad=False):
        z, x = _f32(z, self.device), _f32(x, self.device)
        n = z.shape[0]
        out = torch.empty(n, device=self.device)
        grad = torch.empty((n, self.q), device=self.device) if want_grad else None
        _lib.check(self.lib.bgm_bvn_logpost(self.h, _ptr(z), _ptr(x), n, int(row_base), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                            int(stream_id) & 0xFFFFFFFF, _ptr(out), _ptr(grad), self._stream()), "bgm_bvn_logpost")
        return (out, grad) if want_grad else out

    def hmc_run(self, x, state, logp, grad, step, it_begin, n_iters, burn_in, n_leapfrog, seed, init=False,
                row_base=0, acc_prob=None, acc_count=None, draws=None):
        a = _lib.HmcArgs()
        a.x_dev = x.data_ptr(); a.n = x.shape[0]; a.row_base = int(row_base)
        a.state_dev, a.logp_dev, a.grad_dev = state.data_ptr(), logp.data_ptr(), grad.data_ptr()
        a.init = int(bool(init)); a.it_begin = int(it_begin); a.n_iters = int(n_iters); a.burn_in = int(burn_in)
        a.n_leapfrog = int(n_leapfrog); a.step_dev = step.data_ptr(); a.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        a.acc_prob_sum_dev = acc_prob.data_ptr() if acc_prob is not None else None
        a.acc_count_dev = acc_count.data_ptr() if acc_count is not None else None
        a.draws_dev = draws.data_ptr() if draws is not None else None
        _lib.check(self.lib.bgm_bvn_hmc_run(self.h, C.byref(a), self._stream()), "bgm_bvn_hmc_run")

    def hmc_adapt(self, step, acc_prob, it, n_chains, target=0.75, rate=0.01):
        _lib.check(self.lib.bgm_bgm_hmc_adapt(self.h, _ptr(step), _ptr(acc_prob), int(it), float(n_chains),
                                              float(target), float(rate), self._stream()), "bgm_bgm_hmc_adapt")

    def hmc_sample(self, x, n_mcmc, burn_in, step_size=0.01, n_leapfrog=10, seed=42, row_base=0, n_chains_global=None,
                   reduce_fn=None):
        """tfp_mcmc_sampler (bgm/base.py:709-830) on the stochastic target -> dict(draws, step, acc_count, steps)."""
        dev = self.device
        x = _f32(x, dev)
        n = x.shape[0]
        total = burn_in + n_mcmc
        state = torch.empty((n, self.q), device=dev)
        logp = torch.empty(n, device=dev)
        grad = torch.empty((n, self.q), device=dev)
        step = torch.full((1,), float(step_size), device=dev, dtype=torch.float32)
        acc_prob = torch.zeros(total, device=dev, dtype=torch.float64)
        acc_count = torch.zeros(total, device=dev, dtype=torch.int32)
        draws = torch.empty((n_mcmc, n, self.q), device=dev)
        n_adapt = int(burn_in * 0.8)
        for it in range(n_adapt):
            self.hmc_run(x, state, logp, grad, step, it, 1, burn_in, n_leapfrog, seed, init=(it == 0), row_base=row_base,
                         acc_prob=acc_prob, acc_count=acc_count)
            if reduce_fn is not None:
                reduce_fn(acc_prob[it:it + 1])
            if (n_chains_global or n) > 0:
                self.hmc_adapt(step, acc_prob, it, n_chains_global or n)
        self.hmc_run(x, state, logp, grad, step, n_adapt, total - n_adapt, burn_in, n_leapfrog, seed, init=(n_adapt == 0),
                     row_base=row_base, acc_prob=acc_prob, acc_count=acc_count, draws=draws)
        return dict(draws=draws, step=step, acc_count=acc_count, state=state, logp=logp, grad=grad)

    def decode(self, draws, seed, stream_id, burn_in=0, row_base=0, slot=None, k_slots=0, want_full=False, want_var=False,
               add_noise=True, sign_stride=None, sign_off=0):
        """One generator call over draws [n_draws x n x q] -> (cells | None, full | None[, var]).  The Flipout signs of
        draw d, row r are keyed by d * sign_stride + sign_off + r (default stride: n)."""
        draws = _f32(draws, self.device)
        n_draws, n, _ = draws.shape
        cells = torch.empty((n, k_slots, n_draws), device=self.device) if (slot is not None and k_slots > 0) else None
        full = torch.empty((n_draws, n, self.p), device=self.device) if want_full else None
        var = torch.empty((n_draws, n, self.p), device=self.device) if want_var else None
        _lib.check(self.lib.bgm_bvn_decode(self.h, _ptr(draws), n, int(row_base), n_draws, int(burn_in),
                                           int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream_id) & 0xFFFFFFFF,
                                           int(n if sign_stride is None else sign_stride), int(sign_off),
                                           _ptr(slot) if cells is not None else None,
                                           int(k_slots), _ptr(cells), _ptr(full), _ptr(var), int(bool(add_noise)), self._stream()),
                   "bgm_bvn_decode")
        return (cells, full, var) if want_var else (cells, full)

    def row_mean_quantiles(self, mat, q_lo, q_hi):
        mat = mat.contiguous()
        n_rows, m = mat.shape
        mean = torch.empty(n_rows, device=self.device, dtype=torch.float32)
        lo = torch.empty_like(mean)
        hi = torch.empty_like(mean)
        _lib.check(self.lib.bgm_row_mean_quantiles(self.h, _ptr(mat), n_rows, m, float(q_lo), float(q_hi),
                                                   _ptr(mean), _ptr(lo), _ptr(hi), self._stream()), "bgm_row_mean_quantiles")
        return mean, lo, hi

    # -- EGM warm start (bgm/base.py:190-340) ----------------------------------------------
    def egm_begin(self, batch_size, e_units, dz_units, dx_units, lr, gamma, alpha, e_net, dz, dx):
        """Session on top of the open bvn session: copies its generator; encoder `e_net` ([(W, b)..]), discriminators (dicts)."""
        from .engine import flatten_net, CausalEngine
        cfg = _lib.BgmEgmConfig()
        cfg.batch_size = int(batch_size)
        for name, units in (("e", e_units), ("dz", dz_units), ("dx", dx_units)):
            setattr(cfg, "n_hidden_" + name, len(units))
            arr = getattr(cfg, name + "_units")
            for i, u in enumerate(units):
                arr[i] = int(u)
        cfg.lr, cfg.gamma, cfg.alpha = float(lr), float(gamma), float(alpha)
        te = flatten_net(e_net)
        tz, tx = CausalEngine.flatten_disc(dz), CausalEngine.flatten_disc(dx)
        _lib.check(self.lib.bgm_bvn_egm_begin(self.h, C.byref(cfg), te.ctypes.data_as(C.c_void_p), te.size,
                                              tz.ctypes.data_as(C.c_void_p), tz.size, tx.ctypes.data_as(C.c_void_p), tx.size,
                                              self._stream()), "bgm_bvn_egm_begin")
        self._egm_sizes = (self.n_params, te.size, tz.size, tx.size)
        self.egm_open = True

    def egm_disc_step(self, z, x, noise, eps_z, eps_x, seed, stream_id, apply=True, out=None):
        _lib.check(self.lib.bgm_bvn_egm_disc_step(self.h, _ptr(z), _ptr(x), _ptr(noise), float(eps_z), float(eps_x),
                                                  int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream_id) & 0xFFFFFFFF, int(bool(apply)),
                                                  _ptr(out), self._stream()), "bgm_bvn_egm_disc_step")

    def egm_gen_step(self, z, x, noise1, noise2, seed, stream_id, apply=True, out=None):
        _lib.check(self.lib.bgm_bvn_egm_gen_step(self.h, _ptr(z), _ptr(x), _ptr(noise1), _ptr(noise2), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                 int(stream_id) & 0xFFFFFFFF, int(bool(apply)), _ptr(out), self._stream()),
                   "bgm_bvn_egm_gen_step")

    def egm_read(self, what):
        """0: generator side [g | e], 1: discriminators [dz | dx], 2 / 3: gradients of the last gen / disc step."""
        n_g, n_e, n_dz, n_dx = self._egm_sizes
        buf = np.empty((n_g + n_e) if what in (0, 2) else (n_dz + n_dx), np.float32)
        _lib.check(self.lib.bgm_bvn_egm_read(self.h, int(what), buf.ctypes.data_as(C.c_void_p), buf.size, self._stream()),
                   "bgm_bvn_egm_read")
        return buf

    def egm_write(self, what, buf):
        buf = np.ascontiguousarray(buf, np.float32)
        _lib.check(self.lib.bgm_bvn_egm_write(self.h, int(what), buf.ctypes.data_as(C.c_void_p), buf.size, self._stream()),
                   "bgm_bvn_egm_write")

    def egm_encode(self, x):
        x = _f32(x, self.device)
        z = torch.empty((x.shape[0], self.q), device=self.device)
        _lib.check(self.lib.bgm_bvn_egm_encode(self.h, _ptr(x), x.shape[0], _ptr(z), self._stream()), "bgm_bvn_egm_encode")
        return z

    def egm_sync(self):
        _lib.check(self.lib.bgm_bvn_egm_sync(self.h, self._stream()), "bgm_bvn_egm_sync")

    def egm_end(self):
        _lib.check(self.lib.bgm_bvn_egm_end(self.h, self._stream()), "bgm_bvn_egm_end")
        self.egm_open = False
