"""Python driver of the Bayesian-network (``use_bnn=True``) session of the C ABI (include/bgm_hip.h, bgm_bnn_*).

PyTorch supplies device memory and the HIP stream only; there is no CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .engine import _ptr, _f32, DEFAULT_G_UNITS, DEFAULT_FH_UNITS

NETS = ("g", "e", "f", "h")          # parameter order of a session (BGM_BNN_G, _E, _F, _H)


def flatten_bnn(net):
    """{"gamma", "beta", "layers": [(loc, rho, bias), ...]} -> flat float32 in session order."""
    parts = [net["gamma"], net["beta"]] + [a for L in net["layers"] for a in L]
    return np.concatenate([np.asarray(a, np.float32).ravel() for a in parts]).astype(np.float32)


def unflatten_bnn(theta, dims):
    o = 0

    def take(shape):
        nonlocal o
        n = int(np.prod(shape))
        a = np.array(theta[o:o + n], dtype=np.float32).reshape(shape)
        o += n
        return a
    net = {"gamma": take((dims[0],)), "beta": take((dims[0],)), "layers": []}
    for l in range(len(dims) - 1):
        net["layers"].append((take((dims[l], dims[l + 1])), take((dims[l], dims[l + 1])), take((dims[l + 1],))))
    assert o == len(theta)
    return net


class BnnEngine(object):
    """One session: parameters of g, e, f, h with their Adam slots on the device, step and sampling kernels."""

    def __init__(self, v_dim, z_dims, binary_treatment=False, g_units=None, e_units=None, f_units=None, h_units=None,
                 kl_weight=1e-4, max_batch=32, norm_mode=0, device=0, sigma_v=None, sigma_x=None, sigma_y=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("bayesgm_amd: no HIP device visible; the hot path has no CPU fallback")
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        torch.zeros(1, device=self.device)
        self.v_dim, self.z_dims = int(v_dim), [int(z) for z in z_dims]
        self.q = sum(self.z_dims)
        self.binary = bool(binary_treatment)
        cfg = _lib.BnnConfig()
        cfg.v_dim = self.v_dim
        for i in range(4):
            cfg.z_dims[i] = self.z_dims[i]
        cfg.binary_treatment = int(self.binary)
        units = [list(DEFAULT_G_UNITS if g_units is None else g_units), list(DEFAULT_G_UNITS if e_units is None else e_units),
                 list(DEFAULT_FH_UNITS if f_units is None else f_units), list(DEFAULT_FH_UNITS if h_units is None else h_units)]
        for k in range(4):
            if len(units[k]) + 1 > _lib.BGM_MAX_LAYERS:
                raise ValueError("%s_units: at most %d hidden layers" % (NETS[k], _lib.BGM_MAX_LAYERS - 1))
            cfg.n_hidden[k] = len(units[k])
            for i, u in enumerate(units[k]):
                cfg.units[k][i] = int(u)
        cfg.kl_weight = float(kl_weight)
        cfg.max_batch = int(max_batch)
        cfg.norm_mode = int(norm_mode)          # 0: batch statistics (reference as written), 1: fixed mean 0 / variance 1
        cfg.sigma_v, cfg.sigma_x, cfg.sigma_y = (float(sv) if sv else 0.0 for sv in (sigma_v, sigma_x, sigma_y))     # fixed likelihood sd (0: variance head)
        self.cfg = cfg
        z0, z1, z2, _ = self.z_dims
        ins = [self.q, self.v_dim, z0 + z1 + 1, z0 + z2]
        outs = [self.v_dim + 1, self.q, 2, 2]
        self.dims = {NETS[k]: [ins[k]] + units[k] + [outs[k]] for k in range(4)}
        offs = (C.c_int64 * 5)()
        _lib.check(self.lib.bgm_bnn_layout(C.byref(cfg), offs), "bgm_bnn_layout")
        self.offsets = [int(o) for o in offs]
        self.n_params = self.offsets[4]
        self.h = C.c_void_p()
        _lib.check(self.lib.bgm_create(C.byref(self.h), self.device.index), "bgm_create")
        self.open = False

    def set_disc_norm(self, mode):
        """BatchNormalization mode of the EGM discriminators opened afterwards: "batch" | "fixed" (bgm_set_disc_norm)."""
        _lib.check(self.lib.bgm_set_disc_norm(self.h, {"batch": 0, "fixed": 1}[mode]), "bgm_set_disc_norm")

    def set_precision(self, mode):
        """Arithmetic of logpost / mh_run / effects launched afterwards: "fp32" (default) | "f16x3" (bgm_bnn_set_precision); needs an open
        session and is kept across begin()."""
        self._precision = mode
        if self.open:
            _lib.check(self.lib.bgm_bnn_set_precision(self.h, {"fp32": 0, "bf16x3": 1, "f16x3": 2}[mode]), "bgm_bnn_set_precision")

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

    # -- parameters --------------------------------------------------------------------------------
    def begin(self, model):
        """model: {"g": net, "e": net, "f": net, "h": net} (oracle/bnn.py structure) or a flat float32 vector."""
        theta = model if isinstance(model, np.ndarray) else np.concatenate([flatten_bnn(model[k]) for k in NETS])
        theta = np.ascontiguousarray(theta, np.float32)
        _lib.check(self.lib.bgm_bnn_begin(self.h, C.byref(self.cfg), theta.ctypes.data_as(C.c_void_p), theta.size, self._stream()),
                   "bgm_bnn_begin")
        self.open = True
        if getattr(self, "_precision", "fp32") != "fp32":
            self.set_precision(self._precision)

    MAX_BATCH = 4096      # BNN_MAX_BATCH (csrc/bnn_kernels.h)

    def ensure_max_batch(self, rows):
        """Minibatches of `rows` rows per rank (causalbgm/base.py:434 takes any batch_size): the session's workspaces are sized at
        begin(); a larger minibatch re-opens the session around the same parameters -- possible only before its first step (the
        optimizer clocks start at begin())."""
        rows = int(rows)
        if rows <= self.cfg.max_batch:
            return
        if rows > self.MAX_BATCH:
            raise ValueError("bayesgm_amd: minibatches of Bayesian nets hold at most %d rows per rank (got %d)" % (self.MAX_BATCH, rows))
        if getattr(self, "_stepped", False):
            raise ValueError("bayesgm_amd: batch_size %d exceeds the session's max_batch %d and optimizer steps have been taken; "
                             "create the model with params['max_batch'] >= %d" % (rows, self.cfg.max_batch, rows))
        # (no step taken in this session, but a checkpoint may have restored the Adam slots: they travel with the parameters)
        state = [self.read(w) for w in (0, 2, 3)] if self.open else None
        self.cfg.max_batch = rows
        if state is not None:
            self.begin(state[0])
            if np.any(state[1]) or np.any(state[2]):
                self.write(state[1], 2)
                self.write(state[2], 3)

    def read(self, what=0):
        out = np.empty(self.n_params, np.float32)
        _lib.check(self.lib.bgm_bnn_read(self.h, what, out.ctypes.data_as(C.c_void_p), out.size, self._stream()), "bgm_bnn_read")
        return out

    def write(self, theta, what=0):
        theta = np.ascontiguousarray(theta, np.float32)
        _lib.check(self.lib.bgm_bnn_write(self.h, what, theta.ctypes.data_as(C.c_void_p), theta.size, self._stream()), "bgm_bnn_write")

    def split(self, theta):
        return {NETS[k]: unflatten_bnn(theta[self.offsets[k]:self.offsets[k + 1]], self.dims[NETS[k]]) for k in range(4)}

    def grad_exchange(self, buf, to_session):
        """Copy the session's gradient into `buf` (device tensor [n_params]) or back: the data-parallel all-reduce sits between."""
        _lib.check(self.lib.bgm_bnn_grad_exchange(self.h, _ptr(buf), int(bool(to_session)), self._stream()), "bgm_bnn_grad_exchange")

    def row_mean_quantiles(self, mat, q_lo, q_hi):
        """mat [n_rows, m] on device -> (mean, lo, hi) each [n_rows]."""
        mat = mat.contiguous()
        n_rows, m = mat.shape
        mean = torch.empty(n_rows, device=self.device, dtype=torch.float32)
        lo = torch.empty_like(mean)
        hi = torch.empty_like(mean)
        _lib.check(self.lib.bgm_row_mean_quantiles(self.h, _ptr(mat), n_rows, m, float(q_lo), float(q_hi),
                                                   _ptr(mean), _ptr(lo), _ptr(hi), self._stream()), "bgm_row_mean_quantiles")
        return mean, lo, hi

    # -- minibatch steps -----------------------------------------------------------------------------
    def theta_step(self, data_z, idx, x, y, v, lr_theta, seed, stream_id, apply=True, batch_global=0, out=None):
        self._stepped = True
        _lib.check(self.lib.bgm_bnn_theta_step(self.h, _ptr(data_z), _ptr(idx), _ptr(x), _ptr(y), _ptr(v), idx.numel(),
                                               batch_global, float(lr_theta), int(seed), int(stream_id) & 0xFFFFFFFF,
                                               int(apply), _ptr(out), self._stream()), "bgm_bnn_theta_step")

    def theta_apply(self, lr_theta):
        self._stepped = True
        _lib.check(self.lib.bgm_bnn_theta_apply(self.h, float(lr_theta), self._stream()), "bgm_bnn_theta_apply")

    def z_step(self, x, y, v, data_z, zm, zv, idx, lr_z, seed, stream_id, lazy=False, batch_global=0, out=None, dz_out=None):
        self._stepped = True
        _lib.check(self.lib.bgm_bnn_z_step(self.h, _ptr(x), _ptr(y), _ptr(v), _ptr(data_z), _ptr(zm), _ptr(zv), _ptr(idx),
                                           data_z.shape[0], idx.numel(), batch_global, float(lr_z), int(lazy), int(seed),
                                           int(stream_id) & 0xFFFFFFFF, _ptr(out), _ptr(dz_out), self._stream()),
                   "bgm_bnn_z_step")

    def z_sync(self, data_z, zm, zv, idx, lr_z):
        """Replay mode (lazy = 2): bring the rows idx (None = every row) of the latent table and its Adam slots up to the current step."""
        _lib.check(self.lib.bgm_bnn_z_sync(self.h, _ptr(data_z), _ptr(zm), _ptr(zv), _ptr(idx), data_z.shape[0],
                                           0 if idx is None else int(idx.numel()), float(lr_z), self._stream()), "bgm_bnn_z_sync")

    def fit_epoch(self, x, y, v, data_z, zm, zv, perm, batch, lr_theta, lr_z, lazy, seed, stream_id0, out_t=None, out_z=None):
        """All minibatches perm[0:batch], perm[batch:2 batch], ... of one epoch with the loop inside the library (bgm_bnn_fit_epoch;
        single process).  Returns the number of minibatches run (each consumed three noise streams)."""
        self._stepped = True
        n_done = C.c_int32(0)
        _lib.check(self.lib.bgm_bnn_fit_epoch(self.h, _ptr(x), _ptr(y), _ptr(v), _ptr(data_z), _ptr(zm), _ptr(zv), _ptr(perm),
                                              data_z.shape[0], int(perm.numel()), int(batch), float(lr_theta), float(lr_z), int(lazy),
                                              int(seed), int(stream_id0) & 0xFFFFFFFF, _ptr(out_t), _ptr(out_z), C.byref(n_done),
                                              self._stream()), "bgm_bnn_fit_epoch")
        return n_done.value

    def fit_epoch_dp(self, comm, x, y, v, data_z, zm, zv, perm, batch, lr_theta, lr_z, lazy, seed, stream_id0, out_t=None, out_z=None):
        """This rank's share of a data-parallel epoch (bgm_bnn_fit_epoch_dp): the host loop's steps with ONE ncclAllReduce of the
        session's fused gradient per minibatch, issued from C++.  Returns the number of minibatches run."""
        self._stepped = True
        n_done = C.c_int32(0)
        _lib.check(self.lib.bgm_bnn_fit_epoch_dp(self.h, _ptr(x), _ptr(y), _ptr(v), _ptr(data_z), _ptr(zm), _ptr(zv), _ptr(perm),
                                                 data_z.shape[0], int(perm.numel()), int(batch), float(lr_theta), float(lr_z), int(lazy),
                                                 int(seed), int(stream_id0) & 0xFFFFFFFF, _ptr(out_t), _ptr(out_z), C.byref(n_done),
                                                 comm.handle, self._stream()), "bgm_bnn_fit_epoch_dp")
        return n_done.value

    def serves_block_shares(self):
        """True when the session samples on the default-shape kernels (csrc/bnf_kernels.h), the family that can run a rank's share of one
        block (mh_run(block_row0 > 0)) -- probed through the split-precision switch, which exists for exactly those sessions."""
        prev = getattr(self, "_precision", "fp32")
        if prev == "f16x3":
            return True
        if not self.open or self.lib.bgm_bnn_set_precision(self.h, 2) != 0:
            return False
        _lib.check(self.lib.bgm_bnn_set_precision(self.h, {"fp32": 0, "bf16x3": 1}[prev]), "bgm_bnn_set_precision")
        return True

    # -- large-batch side ------------------------------------------------------------------------------
    def logpost(self, x, y, v, z, block_rows, seed, stream_id, block0=0):
        out = torch.empty(z.shape[0], device=self.device, dtype=torch.float32)
        _lib.check(self.lib.bgm_bnn_logpost(self.h, _ptr(x), _ptr(y), _ptr(v), _ptr(z), z.shape[0], int(block_rows), int(block0),
                                            int(seed), int(stream_id) & 0xFFFFFFFF, _ptr(out), self._stream()), "bgm_bnn_logpost")
        return out

    def mh_run(self, x, y, v, state, block_rows, it_begin, n_iters, burn_in, q_sd, seed, init=False, row_base=0, block0=0,
               acc_count=None, draws=None, n_keep=0, effect=0, sample_y=True, x_values=None, adrf_sum=None, ite=None,
               q_sd_blocks=None, acc_blocks=None, block_row0=0):
        a = _lib.BnnMhArgs()
        a.block_row0 = int(block_row0)      # > 0: the rows are a rank's share of ONE block, starting at this position inside it
        a.x_dev, a.y_dev, a.v_dev = x.data_ptr(), y.data_ptr(), v.data_ptr()
        a.n, a.row_base, a.block_rows, a.block0 = state.shape[0], int(row_base), int(block_rows), int(block0)
        a.state_dev, a.init = state.data_ptr(), int(init)
        a.it_begin, a.n_iters, a.burn_in, a.q_sd, a.seed = int(it_begin), int(n_iters), int(burn_in), float(q_sd), int(seed)
        a.acc_count_dev = acc_count.data_ptr() if acc_count is not None else None
        a.draws_dev = draws.data_ptr() if draws is not None else None
        a.n_keep, a.effect, a.sample_y = int(n_keep), int(effect), int(bool(sample_y))
        a.x_values_dev = x_values.data_ptr() if x_values is not None else None
        a.n_doses = x_values.numel() if x_values is not None else 0
        a.adrf_sum_dev = adrf_sum.data_ptr() if adrf_sum is not None else None
        a.ite_dev = ite.data_ptr() if ite is not None else None
        a.q_sd_blocks_dev = q_sd_blocks.data_ptr() if q_sd_blocks is not None else None
        a.acc_blocks_dev = acc_blocks.data_ptr() if acc_blocks is not None else None
        _lib.check(self.lib.bgm_bnn_mh_run(self.h, C.byref(a), self._stream()), "bgm_bnn_mh_run")

    def effects(self, draws, block_rows, seed, it0=0, x_values=None, sample_y=True, row_base=0, block0=0):
        """infer_from_latent_posterior (base.py:671-763) for draws [n_keep, n, q] on the device: binary -> ITE draws [n_keep, n];
        continuous -> ADRF draws [n_doses, n_keep] (mean over the n rows)."""
        draws = _f32(draws, self.device)
        n_keep, n, _ = draws.shape
        if self.binary:
            ite = torch.empty((n, n_keep), device=self.device, dtype=torch.float32)
            _lib.check(self.lib.bgm_bnn_effects(self.h, _ptr(draws), n, int(block_rows), int(block0), int(row_base), n_keep, int(it0),
                                                int(seed), 2, int(bool(sample_y)), None, 0, None, _ptr(ite), self._stream()), "bgm_bnn_effects")
            return ite.t().contiguous()
        xv = _f32(np.atleast_1d(np.asarray(x_values, dtype=np.float32)), self.device)
        sums = torch.zeros((xv.numel(), n_keep), device=self.device, dtype=torch.float64)
        _lib.check(self.lib.bgm_bnn_effects(self.h, _ptr(draws), n, int(block_rows), int(block0), int(row_base), n_keep, int(it0),
                                            int(seed), 1, int(bool(sample_y)), _ptr(xv), xv.numel(), _ptr(sums), None, self._stream()),
                   "bgm_bnn_effects")
        return (sums / float(n)).float()

    def evaluate(self, x, y, v, z=None, x_values=None, seed=0, stream_id=0, want_sums=True, want_effects=True):
        """evaluate (base.py:534-570) on device tensors; z=None -> z = e(v) (returned).
        Returns (z, sums fp64[3] or None, causal): causal = dose sums fp64 [n_doses] (continuous) / ITE [n] (binary) / None."""
        n = v.shape[0]
        encode = z is None
        if encode:
            z = torch.empty((n, self.q), device=self.device, dtype=torch.float32)
        sums = torch.zeros(3, device=self.device, dtype=torch.float64) if want_sums else None
        dose, ite, xv = None, None, None
        if want_effects:
            if self.binary:
                ite = torch.empty(n, device=self.device, dtype=torch.float32)
            else:
                xv = _f32(np.atleast_1d(np.asarray(x_values, dtype=np.float32)), self.device)
                dose = torch.zeros(xv.numel(), device=self.device, dtype=torch.float64)
        _lib.check(self.lib.bgm_bnn_evaluate(self.h, _ptr(x), _ptr(y), _ptr(v), _ptr(z), int(encode), n, _ptr(xv),
                                             xv.numel() if xv is not None else 0, int(seed), int(stream_id) & 0xFFFFFFFF,
                                             _ptr(sums), _ptr(dose), _ptr(ite), self._stream()), "bgm_bnn_evaluate")
        return z, sums, (ite if self.binary else dose)

    # -- EGM warm start ------------------------------------------------------------------------------
    def egm_begin(self, dz, batch_size, lr, use_z_rec):
        """dz: discriminator dict {"W", "b", "gamma", "beta"} (lists per layer, oracle/egm.py structure)."""
        from .engine import CausalEngine
        theta = np.ascontiguousarray(CausalEngine.flatten_disc(dz), np.float32)
        units = [int(w.shape[1]) for w in dz["W"][:-1]]
        cfg = _lib.EgmConfig()
        cfg.batch_size, cfg.lr, cfg.use_z_rec = int(batch_size), float(lr), int(use_z_rec)
        cfg.n_hidden_dz = len(units)
        for i, u in enumerate(units):
            cfg.dz_units[i] = u
        self.n_dz = theta.size
        self._stepped = True          # (the warm start's state lives in the session: no re-begin from here on, ensure_max_batch)
        _lib.check(self.lib.bgm_bnn_egm_begin(self.h, C.byref(cfg), theta.ctypes.data_as(C.c_void_p), theta.size, self._stream()),
                   "bgm_bnn_egm_begin")

    def egm_disc_step(self, z, idx, v, eps, seed, stream_id, apply=True, out=None):
        _lib.check(self.lib.bgm_bnn_egm_disc_step(self.h, _ptr(z), _ptr(idx), _ptr(v), float(eps), int(seed), int(stream_id) & 0xFFFFFFFF,
                                                  int(apply), _ptr(out), self._stream()), "bgm_bnn_egm_disc_step")

    def egm_gen_step(self, z, idx, v, x, y, seed, stream_id, apply=True, out=None):
        _lib.check(self.lib.bgm_bnn_egm_gen_step(self.h, _ptr(z), _ptr(idx), _ptr(v), _ptr(x), _ptr(y), int(seed),
                                                 int(stream_id) & 0xFFFFFFFF, int(apply), _ptr(out), self._stream()), "bgm_bnn_egm_gen_step")

    def egm_sizes(self):
        return int(self.n_params), int(self.n_dz)

    def egm_set_share(self, row0):
        """This rank's minibatch rows are rows row0 ... of the global minibatch (keys their Flipout sign vectors)."""
        _lib.check(self.lib.bgm_bnn_egm_set_share(self.h, int(row0)), "bgm_bnn_egm_set_share")

    def egm_grad(self, which, scale, out):
        """Data-parallel warm start: the gradient of the last step run with apply=False, times `scale`, into the device tensor `out`."""
        _lib.check(self.lib.bgm_bnn_egm_grad(self.h, int(which), float(scale), _ptr(out), out.numel(), self._stream()), "bgm_bnn_egm_grad")

    def egm_apply(self, which, grad):
        _lib.check(self.lib.bgm_bnn_egm_apply(self.h, int(which), _ptr(grad), grad.numel(), self._stream()), "bgm_bnn_egm_apply")

    def egm_read(self, what):
        out = np.empty(self.n_dz, np.float32)
        _lib.check(self.lib.bgm_bnn_egm_read(self.h, what, out.ctypes.data_as(C.c_void_p), out.size, self._stream()), "bgm_bnn_egm_read")
        return out

    def egm_end(self):
        _lib.check(self.lib.bgm_bnn_egm_end(self.h, self._stream()), "bgm_bnn_egm_end")

    def end(self):
        _lib.check(self.lib.bgm_bnn_end(self.h, self._stream()), "bgm_bnn_end")
        self.open = False
