"""Thin Python driver over the C ABI (include/bgm_hip.h).

PyTorch supplies device memory and the HIP stream only; all arithmetic of the
hot path runs in libbgm_hip.so.  There is no CPU fallback: constructing an
engine without a gfx950 GPU or without the built library raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

DEFAULT_G_UNITS = [64, 64, 64, 64, 64]
DEFAULT_FH_UNITS = [64, 32, 8]


def flatten_net(net):
    """[(W [in,out], b [out]), ...] -> flat float32 in Keras order."""
    return np.concatenate([np.concatenate([np.asarray(W, np.float32).ravel(), np.asarray(b, np.float32).ravel()])
                           for W, b in net]).astype(np.float32)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _f32(t, device):
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
    return t.to(device=device, dtype=torch.float32).contiguous()


class CausalEngine(object):
    """Device-side state of one CausalBGM model: packed weights + kernels."""

    def __init__(self, v_dim, z_dims, binary_treatment=False, g_units=None, f_units=None, h_units=None,
                 e_units=None, sigma_v=None, sigma_x=None, sigma_y=None, device=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("bayesgm_amd: no HIP device visible; the hot path has no CPU fallback")
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        torch.cuda.init()
        torch.zeros(1, device=self.device)  # make sure the HIP context of this device exists
        self.v_dim = int(v_dim)
        self.z_dims = [int(z) for z in z_dims]
        self.q = sum(self.z_dims)
        self.binary = bool(binary_treatment)
        cfg = _lib.CausalConfig()
        cfg.v_dim = self.v_dim
        for i in range(4):
            cfg.z_dims[i] = self.z_dims[i]
        cfg.binary_treatment = int(self.binary)
        for name, units, default in (("g", g_units, DEFAULT_G_UNITS), ("f", f_units, DEFAULT_FH_UNITS),
                                     ("h", h_units, DEFAULT_FH_UNITS), ("e", e_units, DEFAULT_G_UNITS)):
            units = list(default if units is None else units)
            if len(units) > _lib.BGM_MAX_LAYERS:
                raise ValueError("%s_units: at most %d hidden layers" % (name, _lib.BGM_MAX_LAYERS))
            setattr(cfg, "n_hidden_" + name, len(units))
            arr = getattr(cfg, name + "_units")
            for i, u in enumerate(units):
                arr[i] = int(u)
        cfg.sigma_v = float(sigma_v) if sigma_v is not None else -1.0
        cfg.sigma_x = float(sigma_x) if sigma_x is not None else -1.0
        cfg.sigma_y = float(sigma_y) if sigma_y is not None else -1.0
        self.cfg = cfg
        self.h = C.c_void_p()
        _lib.check(self.lib.bgm_create(C.byref(self.h), self.device.index), "bgm_create")
        _lib.check(self.lib.bgm_causal_configure(self.h, C.byref(cfg)), "bgm_causal_configure")

    def set_precision(self, mode):
        """Arithmetic of logpost / mh_run launched afterwards: "fp32" (default) | "bf16x3" | "f16x3" (bgm_causal_set_precision)."""
        _lib.check(self.lib.bgm_causal_set_precision(self.h, {"fp32": 0, "bf16x3": 1, "f16x3": 2}[mode]), "bgm_causal_set_precision")

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

    # -- weights ---------------------------------------------------------
    def set_weights(self, net_id, net):
        theta = flatten_net(net) if not isinstance(net, np.ndarray) else np.ascontiguousarray(net, np.float32)
        _lib.check(self.lib.bgm_causal_set_weights(self.h, net_id, theta.ctypes.data_as(C.c_void_p),
                                                   theta.size, self._stream()), "bgm_causal_set_weights")
        if not hasattr(self, "_w_count"):
            self._w_count = {}
        self._w_count[int(net_id)] = int(theta.size)

    def set_model(self, g=None, f=None, h=None, e=None):
        for nid, net in ((_lib.NET_G, g), (_lib.NET_F, f), (_lib.NET_H, h), (_lib.NET_E, e)):
            if net is not None:
                self.set_weights(nid, net)

    # -- kernels ---------------------------------------------------------
    def logpost(self, x, y, v, z):
        """get_log_posterior (causalbgm/base.py:765-817) -> torch [n] on device."""
        x, y, v, z = (_f32(t, self.device) for t in (x, y, v, z))
        n = v.shape[0]
        out = torch.empty(n, device=self.device, dtype=torch.float32)
        _lib.check(self.lib.bgm_causal_logpost(self.h, _ptr(x), _ptr(y), _ptr(v), _ptr(z), n, _ptr(out),
                                               self._stream()), "bgm_causal_logpost")
        return out

    def encode(self, v):
        v = _f32(v, self.device)
        n = v.shape[0]
        z = torch.empty((n, self.q), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.bgm_causal_encode(self.h, _ptr(v), n, _ptr(z), self._stream()), "bgm_causal_encode")
        return z

    def mh_slots(self, n):
        s = C.c_int32()
        _lib.check(self.lib.bgm_causal_mh_slots(self.h, n, C.byref(s)), "bgm_causal_mh_slots")
        return s.value

    def mh_info(self, n):
        info = _lib.MhInfo()
        _lib.check(self.lib.bgm_causal_mh_info(self.h, n, C.byref(info)), "bgm_causal_mh_info")
        return info

    def mh_run(self, x, y, v, state, logp, it_begin, n_iters, burn_in, q_sd, seed, init=False, row_base=0,
               acc_count=None, draws=None, n_keep=0, effect=_lib.EFFECT_NONE, sample_y=True, x_values=None,
               adrf_partial=None, ite=None, clock=None):
        """One segment of metropolis_hastings_sampler (causalbgm/base.py:860-898) for all rows."""
        a = _lib.MhArgs()
        a.x_dev, a.y_dev, a.v_dev = x.data_ptr(), y.data_ptr(), v.data_ptr()
        a.n = v.shape[0]
        a.row_base = int(row_base)
        a.state_dev, a.logp_dev = state.data_ptr(), logp.data_ptr()
        a.init = int(bool(init))
        a.it_begin, a.n_iters, a.burn_in = int(it_begin), int(n_iters), int(burn_in)
        a.q_sd = float(q_sd)
        a.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        a.acc_count_dev = acc_count.data_ptr() if acc_count is not None else None
        a.draws_dev = draws.data_ptr() if draws is not None else None
        a.n_keep = int(n_keep)
        a.effect = int(effect)
        a.sample_y = int(bool(sample_y))
        a.x_values_dev = x_values.data_ptr() if x_values is not None else None
        a.n_doses = int(x_values.numel()) if x_values is not None else 0
        a.adrf_partial_dev = adrf_partial.data_ptr() if adrf_partial is not None else None
        a.ite_dev = ite.data_ptr() if ite is not None else None
        a.clock_dev = clock.data_ptr() if clock is not None else None
        _lib.check(self.lib.bgm_causal_mh_run(self.h, C.byref(a), self._stream()), "bgm_causal_mh_run")

    def adrf_reduce(self, partial, n_slots, n_doses, n_keep, n_total):
        out = torch.empty((n_doses, n_keep), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.bgm_adrf_reduce(self.h, _ptr(partial), n_slots, n_doses, n_keep, float(n_total),
                                            _ptr(out), self._stream()), "bgm_adrf_reduce")
        return out

    def row_mean_quantiles(self, mat, q_lo, q_hi):
        """mat [n_rows, m] on device -> (mean, lo, hi) each [n_rows]."""
        mat = mat.contiguous()
        n_rows, m = mat.shape
        mean = torch.empty(n_rows, device=self.device, dtype=torch.float32)
        lo = torch.empty_like(mean)
        hi = torch.empty_like(mean)
        _lib.check(self.lib.bgm_row_mean_quantiles(self.h, _ptr(mat), n_rows, m, float(q_lo), float(q_hi),
                                                   _ptr(mean), _ptr(lo), _ptr(hi), self._stream()),
                   "bgm_row_mean_quantiles")
        return mean, lo, hi

    # -- evaluate -----------------------------------------------------------------
    def evaluate(self, x, y, v, z, x_values=None):
        """CausalBGM.evaluate arithmetic (causalbgm/base.py:534-570) on device tensors.
        Returns (sums[3] float64 tensor, causal): causal = ITE [n] (binary) or dose sums [n_doses]."""
        n = v.shape[0]
        sums = torch.zeros(4, device=self.device, dtype=torch.float64)
        if self.binary:
            ite = torch.empty(n, device=self.device, dtype=torch.float32)
            _lib.check(self.lib.bgm_causal_evaluate(self.h, _ptr(x), _ptr(y), _ptr(v), _ptr(z), n, None, 0, _ptr(sums),
                                                    None, _ptr(ite), self._stream()), "bgm_causal_evaluate")
            return sums, ite
        xv = _f32(np.atleast_1d(np.asarray(x_values, dtype=np.float32)), self.device)
        ns = C.c_int32()
        _lib.check(self.lib.bgm_causal_evaluate_slots(self.h, n, C.byref(ns)), "bgm_causal_evaluate_slots")
        partial = torch.zeros((ns.value, xv.numel()), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.bgm_causal_evaluate(self.h, _ptr(x), _ptr(y), _ptr(v), _ptr(z), n, _ptr(xv), xv.numel(),
                                                _ptr(sums), _ptr(partial), None, self._stream()), "bgm_causal_evaluate")
        dose_sums = self.adrf_reduce(partial, ns.value, xv.numel(), 1, 1.0).reshape(-1)
        return sums, dose_sums

    def effects(self, x, draws, burn_in, seed, x_values=None, sample_y=True, row_base=0):
        """infer_from_latent_posterior (causalbgm/base.py:671-763) for draws [n_keep, n, q] on the device:
        binary -> ITE draws [n_keep, n]; continuous -> ADRF draws [n_doses, n_keep] (mean over the n rows)."""
        x = _f32(x, self.device).reshape(-1)
        draws = _f32(draws, self.device)
        n_keep, n, _ = draws.shape
        seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        if self.binary:
            ite = torch.empty((n, n_keep), device=self.device, dtype=torch.float32)
            _lib.check(self.lib.bgm_causal_effects(self.h, _ptr(x), _ptr(draws), n, int(row_base), n_keep, int(burn_in), seed,
                                                   int(bool(sample_y)), None, 0, None, _ptr(ite), self._stream()), "bgm_causal_effects")
            return ite.t().contiguous()
        xv = _f32(np.atleast_1d(np.asarray(x_values, dtype=np.float32)), self.device)
        ns = C.c_int32()
        _lib.check(self.lib.bgm_causal_evaluate_slots(self.h, n, C.byref(ns)), "bgm_causal_evaluate_slots")
        partial = torch.zeros((ns.value, n_keep, xv.numel()), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.bgm_causal_effects(self.h, _ptr(x), _ptr(draws), n, int(row_base), n_keep, int(burn_in), seed,
                                               int(bool(sample_y)), _ptr(xv), xv.numel(), _ptr(partial), None, self._stream()),
                   "bgm_causal_effects")
        return self.adrf_reduce(partial, ns.value, xv.numel(), n_keep, n)

    # -- fit step functions ------------------------------------------------------
    def fit_begin(self, n_rows, max_batch):
        _lib.check(self.lib.bgm_causal_fit_begin(self.h, int(n_rows), int(max_batch), self._stream()),
                   "bgm_causal_fit_begin")
        n = C.c_int64()
        _lib.check(self.lib.bgm_causal_fit_n_params(self.h, C.byref(n)), "bgm_causal_fit_n_params")
        self.n_params = n.value
        return n.value

    def fit_theta_grad(self, x, y, v, data_z, idx, batch_global, grad, loss=None, row_lo=0, batch=None):
        b = int(idx.numel()) if idx is not None else int(batch)
        _lib.check(self.lib.bgm_causal_fit_theta_grad(self.h, _ptr(x), _ptr(y), _ptr(v), _ptr(data_z), _ptr(idx),
                                                      int(row_lo), b, int(batch_global), _ptr(grad), _ptr(loss),
                                                      self._stream()), "bgm_causal_fit_theta_grad")

    def fit_theta_apply(self, grad, lr_theta):
        _lib.check(self.lib.bgm_causal_fit_theta_apply(self.h, _ptr(grad), float(lr_theta), self._stream()),
                   "bgm_causal_fit_theta_apply")

    def fit_z_step(self, x, y, v, data_z, zm, zv, idx, batch_global, lr_z, lazy=False, loss=None):
        """lazy: False / 0 = dense-decay Adam on the whole table, True / 1 = batch rows only, 2 = replay (fit_z_sync first)."""
        _lib.check(self.lib.bgm_causal_fit_z_step(self.h, _ptr(x), _ptr(y), _ptr(v), _ptr(data_z), _ptr(zm), _ptr(zv),
                                                  _ptr(idx), 0, int(idx.numel()), int(batch_global), float(lr_z),
                                                  int(lazy), _ptr(loss), self._stream()),
                   "bgm_causal_fit_z_step")

    def fit_epoch(self, x, y, v, data_z, zm, zv, perm, batch, lr_theta, lr_z, lazy, loss=None, loss_z=None):
        """All minibatches perm[0:batch], perm[batch:2 batch], ... of one epoch with the loop inside the library
        (bgm_causal_fit_epoch; single process)."""
        _lib.check(self.lib.bgm_causal_fit_epoch(self.h, _ptr(x), _ptr(y), _ptr(v), _ptr(data_z), _ptr(zm), _ptr(zv), _ptr(perm),
                                                 int(perm.numel()), int(batch), float(lr_theta), float(lr_z), int(lazy), _ptr(loss),
                                                 _ptr(loss_z), self._stream()), "bgm_causal_fit_epoch")

    def fit_epoch_dp(self, comm, x, y, v, data_z, zm, zv, perm, batch, lr_theta, lr_z, lazy, loss=None, loss_z=None):
        """This rank's share of a data-parallel epoch (bgm_causal_fit_epoch_dp): `batch` local rows per minibatch, the fused g|f|h
        gradient summed over the ranks of `comm` (parallel.DeviceComm) by one ncclAllReduce per step enqueued inside the library."""
        _lib.check(self.lib.bgm_causal_fit_epoch_dp(self.h, _ptr(x), _ptr(y), _ptr(v), _ptr(data_z), _ptr(zm), _ptr(zv), _ptr(perm),
                                                    int(perm.numel()), int(batch), float(lr_theta), float(lr_z), int(lazy), _ptr(loss),
                                                    _ptr(loss_z), comm.handle, self._stream()), "bgm_causal_fit_epoch_dp")

    def describe(self, batch=32):
        """Kernel paths of this handle (sampling; minibatch steps when a fit session is open) as text."""
        buf = C.create_string_buffer(512)
        _lib.check(self.lib.bgm_causal_describe(self.h, int(batch), buf, 512), "bgm_causal_describe")
        return buf.value.decode()

    def fit_z_sync(self, data_z, zm, zv, idx, lr_z):
        """Replay mode: bring the rows idx (None = every row) of the latent table and its Adam slots up to the current step."""
        _lib.check(self.lib.bgm_causal_fit_z_sync(self.h, _ptr(data_z), _ptr(zm), _ptr(zv), _ptr(idx), 0 if idx is None else int(idx.numel()),
                                                  float(lr_z), self._stream()), "bgm_causal_fit_z_sync")

    def fit_z_grad(self, x, y, v, data_z, idx, batch_global, dz_out, loss=None):
        """d(batch-mean negative log joint, standard-normal prior)/d(batch rows of data_z) -> dz_out [batch x q]; no update."""
        _lib.check(self.lib.bgm_causal_fit_z_grad(self.h, _ptr(x), _ptr(y), _ptr(v), _ptr(data_z), _ptr(idx), 0, int(idx.numel()),
                                                  int(batch_global), _ptr(dz_out), _ptr(loss), self._stream()), "bgm_causal_fit_z_grad")

    def set_prior(self, seg=None, tab=None):
        """Conditional latent prior of the sampling calls made afterwards (bgm_causal_set_prior): seg int32 [n] (device), tab float32
        [n_segments x (q + 2)] = mu, 1 / sigma^2, (q / 2) log sigma^2 per segment (device).  None clears it."""
        self._prior_keep = (seg, tab)          # keep the buffers alive while set
        _lib.check(self.lib.bgm_causal_set_prior(self.h, _ptr(seg), _ptr(tab), 0 if tab is None else int(tab.shape[0])),
                   "bgm_causal_set_prior")

    def get_weights(self, net_id, dims):
        """Device parameters of one net -> [(W, b), ...] (Keras order)."""
        count = sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(len(dims) - 1))
        buf = np.empty(count, np.float32)
        _lib.check(self.lib.bgm_causal_get_weights(self.h, int(net_id), buf.ctypes.data_as(C.c_void_p), count,
                                                   self._stream()), "bgm_causal_get_weights")
        out, off = [], 0
        for i in range(len(dims) - 1):
            nw = dims[i] * dims[i + 1]
            out.append((buf[off:off + nw].reshape(dims[i], dims[i + 1]).copy(),
                        buf[off + nw:off + nw + dims[i + 1]].copy()))
            off += nw + dims[i + 1]
        return out

    def fit_state(self, state=None):
        """Optimizer state of the open fit session (bgm_causal_fit_state): read -> dict(m, v, t_theta, t_z); pass such a
        dict to install it."""
        n = self.n_params
        steps = (C.c_int64 * 2)()
        if state is None:
            m, v = np.empty(n, np.float32), np.empty(n, np.float32)
            _lib.check(self.lib.bgm_causal_fit_state(self.h, 0, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), n, steps,
                                                     self._stream()), "bgm_causal_fit_state")
            return dict(m=m, v=v, t_theta=int(steps[0]), t_z=int(steps[1]))
        m = np.ascontiguousarray(state["m"], np.float32)
        v = np.ascontiguousarray(state["v"], np.float32)
        steps[0], steps[1] = int(state["t_theta"]), int(state["t_z"])
        _lib.check(self.lib.bgm_causal_fit_state(self.h, 1, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), n, steps,
                                                 self._stream()), "bgm_causal_fit_state")
        return None

    def fit_end(self):
        _lib.check(self.lib.bgm_causal_fit_end(self.h, self._stream()), "bgm_causal_fit_end")

    # -- EGM warm start (causalbgm/base.py:305-431) -------------------------------
    @staticmethod
    def flatten_disc(dz):
        """{'W','b','gamma','beta'} lists -> flat [W0..WL | b0..bL | gamma.. | beta..] (bgm_hip.h layout)."""
        return np.concatenate([np.asarray(a, np.float32).ravel() for k in ("W", "b", "gamma", "beta") for a in dz[k]])

    def egm_begin(self, batch_size, dz_units, lr, use_z_rec, dz):
        """Start a warm-start session from the installed g, e, f, h and the discriminator `dz`."""
        cfg = _lib.EgmConfig()
        cfg.batch_size = int(batch_size)
        cfg.n_hidden_dz = len(dz_units)
        for i, u in enumerate(dz_units):
            cfg.dz_units[i] = int(u)
        cfg.lr = float(lr)
        cfg.use_z_rec = int(bool(use_z_rec))
        theta = self.flatten_disc(dz) if isinstance(dz, dict) else np.ascontiguousarray(dz, np.float32)
        self._egm_counts = None
        _lib.check(self.lib.bgm_causal_egm_begin(self.h, C.byref(cfg), theta.ctypes.data_as(C.c_void_p), theta.size,
                                                 self._stream()), "bgm_causal_egm_begin")
        self._egm_n_dz = theta.size

    def egm_disc_step(self, z, idx, v, eps, apply=True, out=None):
        _lib.check(self.lib.bgm_causal_egm_disc_step(self.h, _ptr(z), _ptr(idx), _ptr(v), float(eps), int(bool(apply)),
                                                     _ptr(out), self._stream()), "bgm_causal_egm_disc_step")

    def egm_gen_step(self, z, idx, v, x, y, apply=True, out=None):
        _lib.check(self.lib.bgm_causal_egm_gen_step(self.h, _ptr(z), _ptr(idx), _ptr(v), _ptr(x), _ptr(y),
                                                    int(bool(apply)), _ptr(out), self._stream()), "bgm_causal_egm_gen_step")

    def egm_sizes(self):
        """(parameters of g | e | f | h, parameters of the discriminator): the buffer sizes of egm_grad / egm_apply."""
        return sum(self._w_count[k] for k in (_lib.NET_G, _lib.NET_E, _lib.NET_F, _lib.NET_H)), int(self._egm_n_dz)

    def egm_grad(self, which, scale, out):
        """Data-parallel warm start: the gradient of the last step run with apply=False, times `scale`, into the device tensor `out`."""
        _lib.check(self.lib.bgm_causal_egm_grad(self.h, int(which), float(scale), _ptr(out), out.numel(), self._stream()), "bgm_causal_egm_grad")

    def egm_apply(self, which, grad):
        """... and the Adam step of that optimizer from the (all-reduced) gradient."""
        _lib.check(self.lib.bgm_causal_egm_apply(self.h, int(which), _ptr(grad), grad.numel(), self._stream()), "bgm_causal_egm_apply")

    def egm_read(self, what, count):
        """what: 0 generator-side parameters [g|e|f|h], 1 discriminator, 2 / 3 last gen / disc gradients."""
        buf = np.empty(int(count), np.float32)
        _lib.check(self.lib.bgm_causal_egm_read(self.h, int(what), buf.ctypes.data_as(C.c_void_p), buf.size, self._stream()),
                   "bgm_causal_egm_read")
        return buf

    def egm_sync(self):
        _lib.check(self.lib.bgm_causal_egm_sync(self.h, self._stream()), "bgm_causal_egm_sync")

    def egm_end(self):
        _lib.check(self.lib.bgm_causal_egm_end(self.h, self._stream()), "bgm_causal_egm_end")

    OUTCOME_CACHE_NAMES = {"off": 0, "wave": 1, "chain": 2}

    @classmethod
    def outcome_cache_mode(cls, on):
        """bool (False = off, True = per chain), a name ('off' / 'wave' / 'chain') or the C ABI's number (0 / 1 / 2, bgm_hip.h) -> mode.
        (bools are resolved first: True == 1 in Python, but True means 'the default cache' = mode 2, the integer 1 means mode 1.)"""
        if isinstance(on, (bool, np.bool_)):
            return 2 if on else 0
        if isinstance(on, str) and on in cls.OUTCOME_CACHE_NAMES:
            return cls.OUTCOME_CACHE_NAMES[on]
        if isinstance(on, (int, np.integer)) and int(on) in (0, 1, 2):
            return int(on)
        raise ValueError("outcome cache mode must be False / 'off' / 0, 'wave' / 1 or True / 'chain' / 2; got %r" % (on,))

    def set_outcome_cache(self, on=True):
        """Retained phase of the effect samplers: False / 'off' = the outcome net at every retained draw (the reference); 'wave' = reuse
        the (mean, sd) of a 16-chain tile none of whose chains moved; True / 'chain' (default) = per chain, through the event form of the
        retained phase where it exists (csrc/causal_event_kernels.h), else 'wave'.  Bit-identical sums in every mode."""
        _lib.check(self.lib.bgm_causal_set_outcome_cache(self.h, self.outcome_cache_mode(on)), "bgm_causal_set_outcome_cache")

    def set_event_budget(self, n_bytes):
        """Upper bound on the event buffers of one segment of the retained phase in its event form (0: BGM_EVENT_BUDGET_MB or 8 GiB)."""
        _lib.check(self.lib.bgm_causal_set_event_budget(self.h, int(n_bytes)), "bgm_causal_set_event_budget")

    def outcome_cache_stats(self, reset=True):
        """(served, total) since the last reset: retained tile-iterations for the per-wave cache, retained chain-iterations for the
        event form (served = those that needed no outcome-net evaluation)."""
        out = (C.c_int64 * 2)()
        _lib.check(self.lib.bgm_causal_outcome_cache_stats(self.h, out, int(reset)), "bgm_causal_outcome_cache_stats")
        return int(out[0]), int(out[1])

    def timing_enable(self, on=True):
        _lib.check(self.lib.bgm_timing_enable(self.h, int(on)), "bgm_timing_enable")

    def timing_read(self, kind=-1, reset=True):
        """(launches, total ms) of the MH kernel launches of `kind` (EFFECT_*; -1 = all)."""
        n = C.c_int64()
        ms = C.c_double()
        _lib.check(self.lib.bgm_timing_read(self.h, int(kind), C.byref(n), C.byref(ms), int(reset)),
                   "bgm_timing_read")
        return n.value, ms.value

    # -- whole-sampler conveniences -------------------------------------------
    def mh_sample(self, x, y, v, burn_in, n_keep, q_sd, seed, chunk=None, want_draws=False,
                  effect=_lib.EFFECT_NONE, x_values=None, sample_y=True, row_base=0, adaptive=False,
                  initial_q_sd=1.0, target=0.25, tol=0.05, adj_int=50, window=100, acc_reduce=None, n_total=None):
        """metropolis_hastings_sampler (+ fused infer_from_latent_posterior) over all rows.

        Returns dict(state, logp, acc_count [burn_in+n_keep], draws | None, adrf [n_doses,n_keep] | None,
        ite [n, n_keep] | None, q_sd).  acc_reduce / n_total: with the rows of one sampler run sharded over ranks, the all-reduce (sum)
        of the window's acceptance count and the total number of rows, so that every rank adapts the proposal scale identically."""
        dev = self.device
        x, y, v = (_f32(t, dev) for t in (x, y, v))
        x = x.reshape(-1)
        y = y.reshape(-1)
        n = v.shape[0]
        total = burn_in + n_keep
        state = torch.empty((n, self.q), device=dev, dtype=torch.float32)
        logp = torch.empty(n, device=dev, dtype=torch.float32)
        acc = torch.zeros(total, device=dev, dtype=torch.int32)
        draws = torch.empty((n_keep, n, self.q), device=dev, dtype=torch.float32) if want_draws else None
        xv = partial = ite = None
        n_slots = self.mh_slots(n)
        if effect == _lib.EFFECT_ADRF:
            xv = _f32(np.atleast_1d(np.asarray(x_values, dtype=np.float32)), dev)
            partial = torch.zeros((n_slots, n_keep, xv.numel()), device=dev, dtype=torch.float32)
        elif effect == _lib.EFFECT_ITE:
            ite = torch.empty((n, n_keep), device=dev, dtype=torch.float32)
        if adaptive:
            q_sd = initial_q_sd
            chunk = adj_int  # q_sd may change every adj_int iterations during burn-in (base.py:880)
        if chunk is None:
            chunk = total
        it = 0
        while it < total:
            # segment boundaries: the adaptive rule looks at the window *after* iteration it
            # where it % adj_int == 0, i.e. q_sd changes between it and it+1.
            if adaptive and it < burn_in:
                nxt = min(((it // adj_int) + 1) * adj_int + 1, total) if it > 0 else min(adj_int + 1, total)
            else:
                nxt = min(it + (chunk if not adaptive else total), total)
            self.mh_run(x, y, v, state, logp, it, nxt - it, burn_in, q_sd, seed, init=(it == 0),
                        row_base=row_base, acc_count=acc, draws=draws, n_keep=n_keep, effect=effect,
                        sample_y=sample_y, x_values=xv, adrf_partial=partial, ite=ite)
            it = nxt
            last = it - 1  # counter value of the iteration just finished
            if adaptive and last < burn_in and last % adj_int == 0 and last > 0:
                w0 = max(0, last + 1 - window)
                cnt = acc[w0:last + 1].sum().double().reshape(1)
                if acc_reduce is not None:          # rows sharded over ranks: ONE acceptance window over all rows, the same decision everywhere
                    cnt = acc_reduce(cnt)
                rate = float(cnt.item()) / ((last + 1 - w0) * (n if n_total is None else n_total))
                if rate < target - tol:
                    q_sd *= 0.9
                elif rate > target + tol:
                    q_sd *= 1.1
        adrf = None
        if effect == _lib.EFFECT_ADRF:
            adrf = self.adrf_reduce(partial, n_slots, xv.numel(), n_keep, n)
        return dict(state=state, logp=logp, acc_count=acc, draws=draws, adrf=adrf, ite=ite, q_sd=q_sd,
                    adrf_partial=partial, n_slots=n_slots)


def flatten_varnet(g):
    """oracle/reference BaseVariationalNet dict -> flat float32 in the order of bgm_bgm_set_weights."""
    parts = [g["bn"]["gamma"], g["bn"]["beta"], g["bn"]["mean"], g["bn"]["var"]]
    for W, b in g["trunk"]:
        parts += [W, b]
    parts += [g["mean"][0], g["mean"][1], g["var"][0], g["var"][1]]
    return np.concatenate([np.asarray(a, np.float32).ravel() for a in parts]).astype(np.float32)


class BgmEngine(object):
    """Device-side state of one BGM generator (BaseVariationalNet, inference mode) + kernels."""

    def __init__(self, x_dim, z_dim, g_units=None, device=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("bayesgm_amd: no HIP device visible; the hot path has no CPU fallback")
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        torch.cuda.init()
        torch.zeros(1, device=self.device)
        self.p, self.q = int(x_dim), int(z_dim)
        units = list(DEFAULT_G_UNITS if g_units is None else g_units)
        self.units = units
        cfg = _lib.BgmConfig()
        cfg.x_dim, cfg.z_dim, cfg.n_hidden_g = self.p, self.q, len(units)
        for i, u in enumerate(units):
            cfg.g_units[i] = int(u)
        self.h = C.c_void_p()
        _lib.check(self.lib.bgm_create(C.byref(self.h), self.device.index), "bgm_create")
        _lib.check(self.lib.bgm_bgm_configure(self.h, C.byref(cfg)), "bgm_bgm_configure")

    def set_disc_norm(self, mode):
        """BatchNormalization mode of the EGM discriminators opened afterwards: "batch" | "fixed" (bgm_set_disc_norm)."""
        _lib.check(self.lib.bgm_set_disc_norm(self.h, {"batch": 0, "fixed": 1}[mode]), "bgm_set_disc_norm")

    HMC_PRECISIONS = {"fp32": 0, "f16x3": 2}

    def set_precision(self, mode="fp32"):
        """Arithmetic of the generator's products in logpost / hmc_run: "fp32" (default) or "f16x3" (split precision on the
        fp16 matrix instruction, opt-in; bgm_bgm_set_precision)."""
        if mode not in self.HMC_PRECISIONS:
            raise ValueError("hmc precision must be 'fp32' or 'f16x3'; got %r" % (mode,))
        _lib.check(self.lib.bgm_bgm_set_precision(self.h, self.HMC_PRECISIONS[mode]), "bgm_bgm_set_precision")

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

    def set_weights(self, g):
        theta = flatten_varnet(g) if isinstance(g, dict) else np.ascontiguousarray(g, np.float32)
        _lib.check(self.lib.bgm_bgm_set_weights(self.h, theta.ctypes.data_as(C.c_void_p), theta.size, self._stream()),
                   "bgm_bgm_set_weights")

    def logpost(self, z, x, want_grad=False):
        """get_log_posterior (bgm/base.py:665-705); x carries NaN at missing cells."""
        z, x = _f32(z, self.device), _f32(x, self.device)
        n = x.shape[0]
        out = torch.empty(n, device=self.device, dtype=torch.float32)
        grad = torch.empty((n, self.q), device=self.device, dtype=torch.float32) if want_grad else None
        _lib.check(self.lib.bgm_bgm_logpost(self.h, _ptr(z), _ptr(x), n, _ptr(out), _ptr(grad), self._stream()),
                   "bgm_bgm_logpost")
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
        _lib.check(self.lib.bgm_bgm_hmc_run(self.h, C.byref(a), self._stream()), "bgm_bgm_hmc_run")

    def hmc_adapt(self, step, acc_prob, it, n_chains, target=0.75, rate=0.01):
        _lib.check(self.lib.bgm_bgm_hmc_adapt(self.h, _ptr(step), _ptr(acc_prob), int(it), float(n_chains),
                                              float(target), float(rate), self._stream()), "bgm_bgm_hmc_adapt")

    def hmc_sample(self, x, n_mcmc, burn_in, step_size=0.01, n_leapfrog=10, seed=42, row_base=0, want_draws=True,
                   n_chains_global=None, reduce_fn=None):
        """tfp_mcmc_sampler (bgm/base.py:709-830): HMC + SimpleStepSizeAdaptation over int(0.8*burn_in)
        steps, all rows one chain each.  `reduce_fn(tensor)` all-reduces the per-iteration acceptance
        statistic across ranks (the step size is shared by ALL chains)."""
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
        draws = torch.empty((n_mcmc, n, self.q), device=dev) if want_draws else None
        n_adapt = int(burn_in * 0.8)
        n_all = float(n_chains_global if n_chains_global is not None else n)
        it = 0
        while it < n_adapt:          # one launch per transition: the step size changes after each
            self.hmc_run(x, state, logp, grad, step, it, 1, burn_in, n_leapfrog, seed, init=(it == 0),
                         row_base=row_base, acc_prob=acc_prob, acc_count=acc_count, draws=draws)
            if reduce_fn is not None:
                reduce_fn(acc_prob[it:it + 1])
            self.hmc_adapt(step, acc_prob, it, n_all)
            it += 1
        if it < total:
            self.hmc_run(x, state, logp, grad, step, it, total - it, burn_in, n_leapfrog, seed, init=(it == 0),
                         row_base=row_base, acc_prob=acc_prob, acc_count=acc_count, draws=draws)
        return dict(state=state, logp=logp, grad=grad, step=step, acc_prob=acc_prob, acc_count=acc_count, draws=draws)

    def predict_draws(self, draws, burn_in, seed, slot=None, k_slots=0, want_full=False, row_base=0, want_var=False,
                      add_noise=True):
        """predict_on_posteriors (bgm/base.py:511-525) -> (cells [n*k_slots, n_draws] | None, full | None[, var])."""
        n_draws, n, _ = draws.shape
        cells = torch.zeros((n * k_slots, n_draws), device=self.device) if slot is not None else None
        full = torch.empty((n_draws, n, self.p), device=self.device) if want_full else None
        var = torch.empty((n_draws, n, self.p), device=self.device) if want_var else None
        _lib.check(self.lib.bgm_bgm_predict_draws(self.h, _ptr(draws), n, int(row_base), n_draws, int(burn_in),
                                                  int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(slot), int(k_slots),
                                                  _ptr(cells), _ptr(full), _ptr(var), int(bool(add_noise)),
                                                  self._stream()), "bgm_bgm_predict_draws")
        return (cells, full, var) if want_var else (cells, full)

    # -- EGM warm start (bgm/base.py:190-340) ------------------------------------
    def egm_begin(self, batch_size, e_units, dz_units, dx_units, lr, gamma, alpha, e_net, dz, dx):
        """Session from the installed generator, the encoder `e_net` ([(W, b)..]) and the discriminators (dicts)."""
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
        _lib.check(self.lib.bgm_bgm_egm_begin(self.h, C.byref(cfg), te.ctypes.data_as(C.c_void_p), te.size,
                                              tz.ctypes.data_as(C.c_void_p), tz.size, tx.ctypes.data_as(C.c_void_p), tx.size,
                                              self._stream()), "bgm_bgm_egm_begin")
        self._egm_sizes = (self.n_theta(), te.size, tz.size, tx.size)

    def egm_disc_step(self, z, x, noise, eps_z, eps_x, apply=True, out=None):
        _lib.check(self.lib.bgm_bgm_egm_disc_step(self.h, _ptr(z), _ptr(x), _ptr(noise), float(eps_z), float(eps_x),
                                                  int(bool(apply)), _ptr(out), self._stream()), "bgm_bgm_egm_disc_step")

    def egm_gen_step(self, z, x, noise1, noise2, apply=True, out=None):
        _lib.check(self.lib.bgm_bgm_egm_gen_step(self.h, _ptr(z), _ptr(x), _ptr(noise1), _ptr(noise2), int(bool(apply)),
                                                 _ptr(out), self._stream()), "bgm_bgm_egm_gen_step")

    def egm_read(self, what):
        """0: generator side [g | e], 1: discriminators [dz | dx], 2 / 3: gradients of the last gen / disc step."""
        n_g, n_e, n_dz, n_dx = self._egm_sizes
        buf = np.empty((n_g + n_e) if what in (0, 2) else (n_dz + n_dx), np.float32)
        _lib.check(self.lib.bgm_bgm_egm_read(self.h, int(what), buf.ctypes.data_as(C.c_void_p), buf.size, self._stream()),
                   "bgm_bgm_egm_read")
        return buf

    def egm_write(self, what, buf):
        buf = np.ascontiguousarray(buf, np.float32)
        _lib.check(self.lib.bgm_bgm_egm_write(self.h, int(what), buf.ctypes.data_as(C.c_void_p), buf.size, self._stream()),
                   "bgm_bgm_egm_write")

    def egm_encode(self, x):
        x = _f32(x, self.device)
        z = torch.empty((x.shape[0], self.q), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.bgm_bgm_egm_encode(self.h, _ptr(x), x.shape[0], _ptr(z), self._stream()), "bgm_bgm_egm_encode")
        return z

    def egm_sync(self):
        _lib.check(self.lib.bgm_bgm_egm_sync(self.h, self._stream()), "bgm_bgm_egm_sync")

    def egm_end(self):
        _lib.check(self.lib.bgm_bgm_egm_end(self.h, self._stream()), "bgm_bgm_egm_end")

    def row_mean_quantiles(self, mat, q_lo, q_hi):
        mat = mat.contiguous()
        n_rows, m = mat.shape
        mean = torch.empty(n_rows, device=self.device, dtype=torch.float32)
        lo = torch.empty_like(mean)
        hi = torch.empty_like(mean)
        _lib.check(self.lib.bgm_row_mean_quantiles(self.h, _ptr(mat), n_rows, m, float(q_lo), float(q_hi),
                                                   _ptr(mean), _ptr(lo), _ptr(hi), self._stream()),
                   "bgm_row_mean_quantiles")
        return mean, lo, hi

    # -- fit step functions (bgm/base.py:145-187) ---------------------------------
    def fit_begin(self, n_rows, max_batch):
        _lib.check(self.lib.bgm_bgm_fit_begin(self.h, int(n_rows), int(max_batch), self._stream()), "bgm_bgm_fit_begin")
        n = C.c_int64()
        _lib.check(self.lib.bgm_bgm_fit_n_params(self.h, C.byref(n)), "bgm_bgm_fit_n_params")
        self.n_params = n.value
        return n.value

    def fit_set_global_batch(self, batch_global):
        _lib.check(self.lib.bgm_bgm_fit_set_global_batch(self.h, int(batch_global)), "bgm_bgm_fit_set_global_batch")

    def fit_theta_grad(self, x, data_z, idx, grad, loss=None):
        _lib.check(self.lib.bgm_bgm_fit_theta_grad(self.h, _ptr(x), _ptr(data_z), _ptr(idx), int(idx.numel()), _ptr(grad),
                                                   _ptr(loss), self._stream()), "bgm_bgm_fit_theta_grad")

    def fit_theta_apply(self, grad, lr_theta):
        _lib.check(self.lib.bgm_bgm_fit_theta_apply(self.h, _ptr(grad), float(lr_theta), self._stream()),
                   "bgm_bgm_fit_theta_apply")

    def fit_z_step(self, x, data_z, idx, lr_z, loss=None):
        _lib.check(self.lib.bgm_bgm_fit_z_step(self.h, _ptr(x), _ptr(data_z), _ptr(idx), int(idx.numel()), float(lr_z),
                                               _ptr(loss), self._stream()), "bgm_bgm_fit_z_step")

    def fit_epoch(self, x, data_z, perm, n_steps, batch, lr_theta, lr_z, loss=None):
        """The minibatches perm[k * batch : (k + 1) * batch], k < n_steps, in ONE library call (bgm_bgm_fit_epoch)."""
        _lib.check(self.lib.bgm_bgm_fit_epoch(self.h, _ptr(x), _ptr(data_z), _ptr(perm), int(n_steps), int(batch), float(lr_theta),
                                              float(lr_z), _ptr(loss), self._stream()), "bgm_bgm_fit_epoch")

    def get_weights(self):
        """Device parameters -> generator dict (bn / trunk / mean / var)."""
        buf = np.empty(self.n_theta(), np.float32)
        _lib.check(self.lib.bgm_bgm_get_weights(self.h, buf.ctypes.data_as(C.c_void_p), buf.size, self._stream()),
                   "bgm_bgm_get_weights")
        q, p = self.q, self.p
        g = {"bn": {"gamma": buf[:q].copy(), "beta": buf[q:2 * q].copy(), "mean": buf[2 * q:3 * q].copy(),
                    "var": buf[3 * q:4 * q].copy()}, "trunk": []}
        o, d_in = 4 * q, q
        for u in self.units:
            g["trunk"].append((buf[o:o + d_in * u].reshape(d_in, u).copy(), buf[o + d_in * u:o + d_in * u + u].copy()))
            o += d_in * u + u
            d_in = u
        for k in ("mean", "var"):
            g[k] = (buf[o:o + d_in * p].reshape(d_in, p).copy(), buf[o + d_in * p:o + d_in * p + p].copy())
            o += d_in * p + p
        return g

    def n_theta(self):
        n, d_in = 4 * self.q, self.q
        for u in self.units:
            n += d_in * u + u
            d_in = u
        return n + 2 * (d_in * self.p + self.p)

    def fit_end(self):
        _lib.check(self.lib.bgm_bgm_fit_end(self.h, self._stream()), "bgm_bgm_fit_end")
