"""IdentifiableCausalBGM: CausalBGM with the iVAE-style conditional latent prior  Z | U ~ N(mu(U), sigma^2(U) I),  U the one-hot
of a random segment of the row (/root/reference/src/bayesgm/models/causalbgm/identifiable.py:15-616).

What is on the device: everything CausalBGM has (EGM warm start, theta steps, evaluate), the Z-gradient of the minibatch
(`bgm_causal_fit_z_grad`), the prior network (n_segments -> prior_units -> q + 1): its forward / backward, its Adam step and the
fresh-slot Adam step on the batch latents in one small kernel (`bgm_prior_step`, csrc/prior_kernels.h; the reference's
`update_latent_variable_sgd` :150-226, restated in oracle/identifiable.py), the per-segment table of the prior (`bgm_prior_table`)
and the MH / log-posterior kernels that read it (`bgm_causal_set_prior`, the PRIOR = 1 instantiations of csrc/causal_kernels.h).
torch holds the arrays and nothing else.

Stated differences.  (i) The reference's `fit` unpacks seven values from `evaluate`, which returns four (:334 vs base.py:555,570),
so it fails at the first evaluation; the build evaluates as CausalBGM does.  (ii) `use_bnn=True` constructs the Bayesian form,
models/identifiable_bnn.py (a Bayesian prior network on the Bayesian-network kernels).  (iii) `fit` and `predict` shard the rows over
the ranks of torch.distributed (with the adaptive proposal scale the window's acceptance count is all-reduced)."""
import ctypes as C

import numpy as np
import torch

from .. import _lib, parallel
from ..utils import save_data
from .causalbgm import CausalBGM, _init_mlp

def _flatten(net):
    return np.concatenate([np.concatenate([np.asarray(W, np.float32).ravel(), np.asarray(b, np.float32).ravel()]) for W, b in net])


def _unflatten(flat, dims):
    out, o = [], 0
    for i in range(len(dims) - 1):
        W = flat[o:o + dims[i] * dims[i + 1]].reshape(dims[i], dims[i + 1]).copy(); o += dims[i] * dims[i + 1]
        b = flat[o:o + dims[i + 1]].copy(); o += dims[i + 1]
        out.append((W, b))
    return out


class IdentifiableCausalBGM(CausalBGM):
    def __new__(cls, params=None, *args, **kwargs):
        # params['use_bnn'] (indexed without a default by the reference, :56): the Bayesian-network model lives in identifiable_bnn.py
        if cls is IdentifiableCausalBGM and params is not None and params.get('use_bnn', False):
            from .identifiable_bnn import IdentifiableCausalBGMBayes
            inst = object.__new__(IdentifiableCausalBGMBayes)     # (not a subclass of this class: Python will not call __init__ for us)
            inst.__init__(params, *args, **kwargs)
            return inst
        return object.__new__(cls)

    def __init__(self, params, timestamp=None, random_seed=None, device=None):
        if 'n_segments' not in params:
            params['n_segments'] = 10                                                         # :50-51
        CausalBGM.__init__(self, params, timestamp=timestamp, random_seed=random_seed, device=device)
        q = self.engine.q
        dims = [int(params['n_segments'])] + list(params.get('prior_units', [64])) + [q + 1]          # :76-78
        dev = self.engine.device
        if len(dims) - 1 > 4:
            raise NotImplementedError("bayesgm_amd: prior_units with more than 3 hidden layers")
        self._prior_dims = dims
        self._prior_cfg = _lib.PriorConfig(len(dims) - 1, (C.c_int32 * 5)(*(dims + [0] * (5 - len(dims)))))
        self._prior_theta = torch.from_numpy(_flatten(_init_mlp(self._rs, dims))).to(dev)       # Keras order: W0, b0, W1, b1, ...
        self._prior_m = torch.zeros_like(self._prior_theta)
        self._prior_v = torch.zeros_like(self._prior_theta)
        self._prior_t = 0
        self._z_t = 0
        if getattr(self, "_pending_prior", None) is not None:      # a checkpoint was restored by CausalBGM.__init__ before the prior net existed
            self._apply_prior_arrays(self._pending_prior)
            self._pending_prior = None

    # ------------------------------------------------------------------ checkpoints: prior_net and prior_optimizer are tracked too (:112-128)
    def _checkpoint_extra(self):
        flat = {}
        for name, t in (("", self._prior_theta), ("m", self._prior_m), ("v", self._prior_v)):
            for i, (W, b) in enumerate(_unflatten(t.cpu().numpy(), self._prior_dims)):
                flat["prior_%sW%d" % (name, i)] = W
                flat["prior_%sb%d" % (name, i)] = b
        flat["prior_steps"] = np.array([self._prior_t, self._z_t], np.int64)
        if getattr(self, "segments", None) is not None:
            flat["segments"] = np.asarray(self.segments, np.int64)
        return flat

    def _apply_prior_arrays(self, d):
        dev = self.engine.device
        n = len(self._prior_dims) - 1
        if any(("prior_W%d" % i) not in d for i in range(n)):
            return
        T = lambda pre: torch.from_numpy(_flatten([(d["prior_%sW%d" % (pre, i)], d["prior_%sb%d" % (pre, i)]) for i in range(n)])).to(dev)
        self._prior_theta = T("")
        if "prior_mW0" in d:
            self._prior_m, self._prior_v = T("m"), T("v")
            self._prior_t, self._z_t = int(d["prior_steps"][0]), int(d["prior_steps"][1])
        if "segments" in d:
            self.segments = np.asarray(d["segments"])

    def _restore_extra(self, d):
        arrays = {k: d[k] for k in d.files if k.startswith("prior_") or k == "segments"}
        if hasattr(self, "_prior_theta"):
            self._apply_prior_arrays(arrays)
        else:
            self._pending_prior = arrays

    # ------------------------------------------------------------------ prior network
    def prior_parameters(self):
        """[(W, b), ...] of the prior network as NumPy arrays (Keras order)."""
        return _unflatten(self._prior_theta.cpu().numpy(), self._prior_dims)

    def set_prior_parameters(self, net):
        self._prior_theta = torch.from_numpy(_flatten(net)).to(self.engine.device)

    def _prior_table(self):
        """Per segment: mu [q], 1 / sigma^2, (q / 2) log sigma^2 -- what the sampling kernels read (bgm_causal_set_prior)."""
        eng = self.engine
        tab = torch.empty((int(self.params['n_segments']), eng.q + 2), device=eng.device, dtype=torch.float32)
        _lib.check(eng.lib.bgm_prior_table(eng.h, C.byref(self._prior_cfg), self._prior_theta.data_ptr(), tab.data_ptr(), eng._stream()),
                   "bgm_prior_table")
        return tab

    def _with_prior(self, seg_dev):
        self.engine.set_prior(seg_dev.to(torch.int32).contiguous(), self._prior_table())

    # ------------------------------------------------------------------ latent / prior step (:150-226)
    def _z_and_prior_step(self, x, y, v, idx, seg_dev, lr_z, lr_theta, dz, loss_z, out):
        """NLL gradients of the batch latents (bgm_causal_fit_z_grad), then the joint latent / prior-net step (bgm_prior_step);
        out [2] receives the batch means of the conditional-prior term and of |z|^2 / 2."""
        eng = self.engine
        B = int(idx.numel())
        world = parallel.world_size()
        if world > 1:
            # data parallel: the batch means run over the rows of all ranks; the latent step is local, the prior net's gradient is
            # all-reduced before its Adam step, so every rank holds the same prior net
            bg = B * world
            eng.fit_z_grad(x, y, v, self.data_z, idx, bg, dz, loss_z)
            self._z_t += 1
            self._prior_t += 1
            if getattr(self, "_prior_grad", None) is None or self._prior_grad.numel() != self._prior_theta.numel():
                self._prior_grad = torch.empty_like(self._prior_theta)
            _lib.check(eng.lib.bgm_prior_grad(eng.h, C.byref(self._prior_cfg), self._prior_theta.data_ptr(), seg_dev.data_ptr(),
                                              self.data_z.data_ptr(), idx.data_ptr(), B, bg, dz.data_ptr(), float(lr_z), self._z_t,
                                              self._prior_grad.data_ptr(), out.data_ptr(), eng._stream()), "bgm_prior_grad")
            parallel.all_reduce_sum_(self._prior_grad)
            _lib.check(eng.lib.bgm_prior_apply(eng.h, C.byref(self._prior_cfg), self._prior_theta.data_ptr(), self._prior_m.data_ptr(),
                                               self._prior_v.data_ptr(), self._prior_grad.data_ptr(), float(lr_theta), self._prior_t,
                                               eng._stream()), "bgm_prior_apply")
            return
        eng.fit_z_grad(x, y, v, self.data_z, idx, B, dz, loss_z)                      # NLL terms + z / B (standard prior)
        self._z_t += 1
        self._prior_t += 1
        _lib.check(eng.lib.bgm_prior_step(eng.h, C.byref(self._prior_cfg), self._prior_theta.data_ptr(), self._prior_m.data_ptr(),
                                          self._prior_v.data_ptr(), seg_dev.data_ptr(), self.data_z.data_ptr(), idx.data_ptr(), B,
                                          dz.data_ptr(), float(lr_z), float(lr_theta), self._z_t, self._prior_t, out.data_ptr(),
                                          eng._stream()), "bgm_prior_step")

    # ------------------------------------------------------------------ fit (:228-346)
    def fit(self, data, batch_size=32, epochs=100, epochs_per_eval=5, startoff=0, use_egm_init=True, egm_n_iter=30000,
            egm_batches_per_eval=500, verbose=1, save_format='txt'):
        # Under torch.distributed: rows (and their segments, latents) are sharded, batch_size is the GLOBAL minibatch, the fused g | f | h
        # gradient and the prior net's gradient are all-reduced before their Adam steps (all ranks hold identical networks).
        data_x, data_y, data_v = data
        n_total = len(data_x)
        world = parallel.world_size()
        lo_r, hi_r = parallel.shard_range(n_total)
        n = hi_r - lo_r
        b_loc = max(1, batch_size // world)
        n_use = n_total // world if world > 1 else n             # every rank takes the same number of steps (one all-reduce per step)
        eng, dev, q = self.engine, self.engine.device, self.engine.q
        k = int(self.params['n_segments'])
        if verbose:
            print(f"Generating auxiliary variable U for {k} segments.")
        self.segments = np.random.randint(0, k, size=n_total)                                               # :283 (the shared host stream: same draw on every rank)
        seg_dev = torch.from_numpy(self.segments[lo_r:hi_r].astype(np.int32)).to(dev)
        if self._p['save_res'] and parallel.rank() == 0:
            with open('{}/params.txt'.format(self.save_dir), 'w') as f_params:
                f_params.write(str(self.params))
        if use_egm_init:
            self.egm_init(data, egm_n_iter=egm_n_iter, egm_batches_per_eval=egm_batches_per_eval, batch_size=batch_size, verbose=verbose)
        x = self._dev(data_x[lo_r:hi_r]).reshape(-1)
        y = self._dev(data_y[lo_r:hi_r]).reshape(-1)
        v = self._dev(data_v[lo_r:hi_r])
        if use_egm_init:
            if verbose:
                print('Initialize latent variables Z with e(V)...')
            self.data_z = eng.encode(v)
        else:
            if verbose:
                print('Random initialization of latent variables Z...')
            self.data_z = self._dev(np.random.normal(0, 1, size=(n_total, q)).astype('float32')[lo_r:hi_r])
        batch_size = b_loc
        n_params = eng.fit_begin(n, batch_size)
        grad = torch.empty(n_params, device=dev)
        dz = torch.empty((batch_size, q), device=dev)
        loss = torch.zeros(8, device=dev, dtype=torch.float64)
        loss_z = torch.zeros(8, device=dev, dtype=torch.float64)
        step_out = torch.zeros(2, device=dev)
        prior_acc = torch.zeros(2, device=dev, dtype=torch.float64)
        best_loss = np.inf
        self.fit_history = []
        if verbose:
            print('Iterative Updating Starts ...')
        try:
            for epoch in range(epochs + 1):
                sample_idx = torch.from_numpy(np.random.choice(n, n, replace=False).astype(np.int32)).to(dev)
                loss.zero_()
                loss_z.zero_()
                n_rows = 0
                prior_acc.zero_()
                for i in range(0, n_use - batch_size + 1, batch_size):                                      # incomplete last batch skipped (:299)
                    idx = sample_idx[i:i + batch_size]
                    eng.fit_theta_grad(x, y, v, self.data_z, idx, batch_size * world, grad, loss)
                    parallel.all_reduce_sum_(grad)                                                          # fused g | f | h gradient
                    eng.fit_theta_apply(grad, self._p['lr_theta'])
                    self._z_and_prior_step(x, y, v, idx, seg_dev, self._p['lr_z'], self._p['lr_theta'], dz, loss_z, step_out)
                    prior_acc += step_out                                                       # (device-side: no sync per minibatch)
                    n_rows += batch_size
                if world > 1:                        # epoch statistics over all ranks (the step outputs are shares of global batch means)
                    for t_ in (loss, loss_z, prior_acc):
                        parallel.all_reduce_sum_(t_)
                    n_rows *= world
                prior_sum, std_sum = (float(a) * batch_size * world for a in prior_acc.cpu().numpy())
                l = loss.cpu().numpy() / max(1, n_rows)
                lz = loss_z.cpu().numpy() / max(1, n_rows)
                post = float(lz[6]) + (prior_sum - std_sum) / max(1, n_rows)        # kernel sum carries |z|^2 / 2: exchange the prior term
                self.fit_history.append(dict(epoch=epoch, loss_v=float(l[0]), loss_mse_v=float(l[1] / eng.v_dim), loss_x=float(l[2]),
                                             loss_mse_x=float(l[3]), loss_y=float(l[4]), loss_mse_y=float(l[5]), loss_postrior_z=post))
                if verbose:
                    print('Epoch [%d/%d]: loss_px_z [%.4f], loss_mse_x [%.4f], loss_py_z [%.4f], loss_mse_y [%.4f], loss_pv_z [%.4f], '
                          'loss_mse_v [%.4f], loss_postrior_z [%.4f]' % (epoch, epochs, l[2], l[3], l[4], l[5], l[0], l[1] / eng.v_dim, post))
                if epoch % epochs_per_eval == 0:
                    causal_pre, mse_x, mse_y, mse_v = self._evaluate_dev(x, y, v, self.data_z, n_total, lo_r)
                    self.fit_history[-1].update(mse_x=float(mse_x), mse_y=float(mse_y), mse_v=float(mse_v))
                    if verbose:
                        print('Epoch [%d/%d]: MSE_x: %.4f, MSE_y: %.4f, MSE_v: %.4f\n' % (epoch, epochs, mse_x, mse_y, mse_v))
                    if epoch >= startoff and mse_y < best_loss:
                        best_loss = mse_y
                        self.best_causal_pre = causal_pre
                        self.best_epoch = epoch
                    if self._p['save_res'] and parallel.rank() == 0:
                        save_data('{}/causal_pre_at_{}.{}'.format(self.save_dir, epoch, save_format), causal_pre)
        finally:
            eng.fit_end()
            self._pull_weights()

    # ------------------------------------------------------------------ sampling (:521-614)
    def _segments_for(self, n, data_u=None):
        k = int(self.params['n_segments'])
        if data_u is None:
            return np.random.randint(0, k, size=n)                                                # fresh U at predict time (:563-564)
        return np.asarray(data_u).argmax(axis=1)

    def get_log_posterior(self, data_x, data_y, data_v, data_z, data_u, eps=1e-6):
        """log p(z | x, y, v, u) + const, shape (n,) (:521-555); data_u one-hot [n, n_segments]."""
        seg = torch.from_numpy(self._segments_for(len(data_x), data_u).astype(np.int32)).to(self.engine.device)
        self._with_prior(seg)
        try:
            return CausalBGM.get_log_posterior(self, data_x, data_y, data_v, data_z)
        finally:
            self.engine.set_prior(None, None)

    def metropolis_hastings_sampler(self, data, initial_q_sd=1.0, q_sd=None, burn_in=5000, n_keep=3000, target_acceptance_rate=0.25,
                                    tolerance=0.05, adjustment_interval=50, adaptive_sd=None, window_size=100):
        """(samples [n_keep, n, q], data_u one-hot [n, n_segments]) (:557-614)."""
        data_x = data[0]
        segs = self._segments_for(len(data_x))
        self._with_prior(torch.from_numpy(segs.astype(np.int32)).to(self.engine.device))
        try:
            samples = CausalBGM.metropolis_hastings_sampler(self, data, initial_q_sd=initial_q_sd, q_sd=q_sd, burn_in=burn_in, n_keep=n_keep,
                                                            target_acceptance_rate=target_acceptance_rate, tolerance=tolerance,
                                                            adjustment_interval=adjustment_interval, adaptive_sd=adaptive_sd,
                                                            window_size=window_size)
        finally:
            self.engine.set_prior(None, None)
        return samples, np.eye(int(self.params['n_segments']), dtype=np.float32)[segs]

    def predict(self, data, alpha=0.01, n_mcmc=3000, x_values=None, q_sd=1.0, sample_y=True, bs=100, burn_in=5000, verbose=1):
        """Causal effects with posterior intervals (:348-420): one MH run over all rows with a fresh random U (burn-in 5000, the
        sampler's default there), effects fused into the sampling kernel; `bs` only chunked the host-side effect pass of the
        reference and does not change the result."""
        assert 0 < alpha < 1, "The significance level 'alpha' must be greater than 0 and less than 1."
        parallel.check_n_mcmc(n_mcmc)
        binary = bool(self._p['binary_treatment'])
        if not binary and x_values is None:
            raise ValueError("For continuous treatment, 'x_values' must not be None.")
        if x_values is not None:
            x_values = np.array([x_values], dtype=float) if np.isscalar(x_values) else np.array(x_values, dtype=float)
        data_x, data_y, data_v = data
        n = len(data_x)
        eng = self.engine
        if verbose:
            print('MCMC Latent Variable Sampling ...')
        adaptive = (q_sd is None) or (q_sd <= 0)
        # fresh U for all rows (:563-564), drawn once: rank 0's draw is everybody's
        seg_all = parallel.broadcast_(torch.from_numpy(self._segments_for(n).astype(np.int32)).to(eng.device))
        lo_r, hi_r = parallel.shard_range(n)                       # chains are keyed by the global row; an adaptive proposal scale uses ONE
                                                                   # acceptance window over all rows (:585-606): its count is all-reduced
        tab = self._prior_table()
        eng.set_prior(seg_all[lo_r:hi_r].contiguous(), tab)
        try:
            out = eng.mh_sample(self._dev(data_x[lo_r:hi_r]).reshape(-1), self._dev(data_y[lo_r:hi_r]).reshape(-1), self._dev(data_v[lo_r:hi_r]),
                                burn_in, n_mcmc, q_sd, self._next_seed(), effect=_lib.EFFECT_ITE if binary else _lib.EFFECT_ADRF,
                                x_values=x_values, sample_y=sample_y, adaptive=adaptive, row_base=lo_r,
                                acc_reduce=parallel.all_reduce_sum_ if (adaptive and parallel.is_dist()) else None, n_total=n)
        finally:
            eng.set_prior(None, None)
        total = burn_in + n_mcmc
        self._report_acceptance(float(out["acc_count"][max(0, total - 100):].sum().item()), min(100, total), n, verbose)
        if binary:
            mean, lo, hi = eng.row_mean_quantiles(out["ite"], alpha / 2, 1 - alpha / 2)
            res = torch.zeros((3, n), device=eng.device, dtype=torch.float32)
            res[0, lo_r:hi_r], res[1, lo_r:hi_r], res[2, lo_r:hi_r] = mean, lo, hi
            res = parallel.all_reduce_sum_(res).cpu().numpy()          # disjoint row sets: the sum is the gather
            return res[0], np.stack([res[1], res[2]], axis=1)
        sums = parallel.all_reduce_sum_(out["adrf"].double() * float(hi_r - lo_r))       # [n_doses x n_mcmc] draw sums over all rows
        eff = (sums / float(n)).float().contiguous()
        mean, lo, hi = eng.row_mean_quantiles(eff, alpha / 2, 1 - alpha / 2)
        return mean.cpu().numpy(), torch.stack([lo, hi], dim=1).cpu().numpy()
