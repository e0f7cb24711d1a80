"""IdentifiableCausalBGM with Bayesian networks (params['use_bnn'] = True; the reference's default in every causal YAML):
CausalBGM's Bayesian g, e, f, h (causalbgm_bnn.py) plus a BAYESIAN prior network  Z | U ~ N(mu(U), sigma^2(U) I),
prior_net = BayesianFullyConnectedNet(n_segments -> prior_units -> q + 1)
(/root/reference/src/bayesgm/models/causalbgm/identifiable.py:56-67).

On the device: everything CausalBGMBayes has (EGM warm start, theta steps, evaluate, effects); the NLL gradient of the batch latents
(`bgm_bnn_z_step` with dz_out: every net called twice with independent noise, base.py:246-302); the joint latent / prior-net step
(`bgm_bprior_step`, csrc/bprior_kernels.h: conditional-prior term + kl_weight * sum(prior_net.losses), fresh-slot Adam on the batch
latents, prior_optimizer on the prior net; identifiable.py:195-226); one noisy call of the prior net per log-posterior evaluation
inside the Metropolis-Hastings sampler (`bgm_bnn_set_prior`, :541-551).  oracle: oracle/identifiable.py (bnn_* functions).

Stated differences.  (i) As for the deterministic form, the reference's `fit` fails at its first evaluation (unpacks seven of
`evaluate`'s four values); the build evaluates as CausalBGM does.  (ii) params['bnn_norm'] = 'fixed' (the build's default) or 'batch' (the reference as written: every input BatchNormalization, the prior
net's on the one-hot segments included, normalises with the statistics of the block of rows of the call; the sampler then runs on the
any-width path, csrc/bnw_kernels.h, which makes the statistics passes, replicated under torch.distributed); any hidden widths (outside the
default shapes the any-width path reads the same per-row prior tables).  (iii) `predict` treats the panel as ONE block, as the reference's sampler call does (:397); under
torch.distributed every rank runs it on the whole panel (same result everywhere), `fit` shards rows and all-reduces the gradients."""
import ctypes as C

import numpy as np
import torch

from .. import _lib, parallel
from ..bnn_engine import flatten_bnn, unflatten_bnn
from ..utils import save_data
from .causalbgm_bnn import CausalBGMBayes, _init_bnn


class IdentifiableCausalBGMBayes(CausalBGMBayes):
    def __init__(self, params, timestamp=None, random_seed=None, device=None):
        if 'n_segments' not in params:
            params['n_segments'] = 10                                                         # :50-51
        CausalBGMBayes.__init__(self, params, timestamp=timestamp, random_seed=random_seed, device=device)
        q = self.engine.q
        dims = [int(params['n_segments'])] + list(params.get('prior_units', [64])) + [q + 1]          # :66-67
        if len(dims) - 1 > 4:
            raise NotImplementedError("bayesgm_amd: prior_units with more than 3 hidden layers")
        self._prior_dims = dims
        self._prior_cfg = _lib.PriorConfig(len(dims) - 1, (C.c_int32 * 5)(*(dims + [0] * (5 - len(dims)))))
        dev = self.engine.device
        self._prior_theta = torch.from_numpy(flatten_bnn(_init_bnn(self._rs, dims))).to(dev)
        self._prior_m = torch.zeros_like(self._prior_theta)
        self._prior_v = torch.zeros_like(self._prior_theta)
        self._prior_t = 0
        self._z_t = 0
        self._norm_mode = {"batch": 0, "fixed": 1}[self._bnn_norm]
        self._prior_klw = float(params.get('kl_weight', 1.0))                                  # :214
        if getattr(self, "_pending_prior", None) is not None:
            self._apply_prior_arrays(self._pending_prior)
            self._pending_prior = None

    # ------------------------------------------------------------------ prior network
    def prior_parameters(self):
        """{"gamma", "beta", "layers": [(loc, rho, bias), ...]} of the prior network as NumPy arrays."""
        return unflatten_bnn(self._prior_theta.cpu().numpy(), self._prior_dims)

    def set_prior_parameters(self, net):
        self._prior_theta = torch.from_numpy(flatten_bnn(net)).to(self.engine.device)

    def _set_prior(self, seg_dev):
        eng = self.engine
        self._prior_keep = (seg_dev, self._prior_theta)
        _lib.check(eng.lib.bgm_bnn_set_prior(eng.h, C.byref(self._prior_cfg), self._prior_theta.data_ptr(), seg_dev.data_ptr()), "bgm_bnn_set_prior")

    def _clear_prior(self):
        eng = self.engine
        _lib.check(eng.lib.bgm_bnn_set_prior(eng.h, C.byref(self._prior_cfg), None, None), "bgm_bnn_set_prior")
        self._prior_keep = None

    # ------------------------------------------------------------------ checkpoints: prior_net and prior_optimizer are tracked too (:112-128)
    def save_checkpoint(self, epoch):
        flat = dict(theta=self.engine.read(0), opt_m=self.engine.read(2), opt_v=self.engine.read(3), stream=np.array([self._stream], np.int64),
                    prior_theta=self._prior_theta.cpu().numpy(), prior_m=self._prior_m.cpu().numpy(), prior_v=self._prior_v.cpu().numpy(),
                    prior_steps=np.array([self._prior_t, self._z_t], np.int64))
        if self.data_z is not None:
            flat["data_z"] = self.data_z.cpu().numpy()
        if getattr(self, "segments", None) is not None:
            flat["segments"] = np.asarray(self.segments, np.int64)
        path = self.ckpt_manager.save("ckpt-%s.npz" % epoch, flat)
        print('Saving checkpoint for epoch {} at {}'.format(epoch, path))
        return path

    def _apply_prior_arrays(self, d):
        dev = self.engine.device
        if "prior_theta" in d and d["prior_theta"].size == self._prior_theta.numel():
            self._prior_theta = torch.from_numpy(np.asarray(d["prior_theta"], np.float32)).to(dev)
            self._prior_m = torch.from_numpy(np.asarray(d["prior_m"], np.float32)).to(dev)
            self._prior_v = torch.from_numpy(np.asarray(d["prior_v"], np.float32)).to(dev)
            self._prior_t, self._z_t = int(d["prior_steps"][0]), int(d["prior_steps"][1])
        if "segments" in d:
            self.segments = np.asarray(d["segments"])

    def load_checkpoint(self, path):
        CausalBGMBayes.load_checkpoint(self, path)
        d = np.load(path)
        arrays = {k: d[k] for k in d.files if k.startswith("prior_") or k == "segments"}
        if hasattr(self, "_prior_theta"):
            self._apply_prior_arrays(arrays)
        else:
            self._pending_prior = arrays

    # ------------------------------------------------------------------ fit (:228-346)
    def fit(self, data, batch_size=32, epochs=100, epochs_per_eval=5, startoff=0, use_egm_init=True, egm_n_iter=30000,
            egm_batches_per_eval=500, verbose=1, save_format='txt'):
        # Under torch.distributed: rows (their segments, latents) are sharded, batch_size is the GLOBAL minibatch, the g | h | f gradient and
        # the data part of the prior net's gradient are all-reduced before their Adam steps (all ranks hold identical networks).
        data_x, data_y, data_v = data
        n_total = len(data_x)
        world = parallel.world_size()
        lo_r, hi_r = parallel.shard_range(n_total)
        n = hi_r - lo_r
        b_loc = max(2, batch_size // world)
        n_use = n_total // world if world > 1 else n
        eng, dev, q = self.engine, self.engine.device, self.engine.q
        k = int(self.params['n_segments'])
        if verbose:
            print(f"Generating auxiliary variable U for {k} segments.")
        self.segments = np.random.randint(0, k, size=n_total)                                               # :283
        if world > 1:       # one draw for the job: rank 0's (every rank trains on, and rank 0 checkpoints, the same U)
            seg_all = torch.from_numpy(self.segments.astype(np.int32)).to(dev)
            parallel.broadcast_(seg_all, 0)
            self.segments = seg_all.cpu().numpy().astype(np.int64)
        seg_dev = torch.from_numpy(self.segments[lo_r:hi_r].astype(np.int32)).to(dev)
        if self._p['save_res'] and parallel.rank() == 0:
            with open('{}/params.txt'.format(self.save_dir), 'w') as f_params:
                f_params.write(str(self.params))
        if use_egm_init:
            self.egm_init(data, egm_n_iter=egm_n_iter, egm_batches_per_eval=egm_batches_per_eval, batch_size=batch_size, verbose=verbose)
        x = self._dev(data_x[lo_r:hi_r]).reshape(-1)
        y = self._dev(data_y[lo_r:hi_r]).reshape(-1)
        v = self._dev(data_v[lo_r:hi_r])
        eng.ensure_max_batch(b_loc)          # (any batch_size, also without the warm start that would have sized the session)
        seed = self._noise_seed(per_rank=True)
        seed_shared = self._noise_seed(per_rank=False)      # the prior net's perturbation is ONE per global minibatch (Flipout), as in the EGM steps
        if use_egm_init:
            if verbose:
                print('Initialize latent variables Z with e(V)...')
            self.data_z, _, _ = eng.evaluate(None, None, v, None, seed=seed, stream_id=self._streams(1), want_sums=False, want_effects=False)
        else:
            if verbose:
                print('Random initialization of latent variables Z...')
            self.data_z = self._dev(np.random.normal(0, 1, size=(n_total, q)).astype('float32')[lo_r:hi_r])
        grad = torch.empty(eng.n_params, device=dev, dtype=torch.float32) if world > 1 else None
        pgrad = torch.empty_like(self._prior_theta) if world > 1 else None
        dz = torch.empty((b_loc, q), device=dev)
        out_t = torch.zeros(8, device=dev)
        out_z = torch.zeros(4, device=dev)
        out_p = torch.zeros(3, device=dev)
        acc = torch.zeros(4, device=dev, dtype=torch.float64)      # epoch sums: NLL + |z|^2 / 2 loss, prior term, |z|^2 / 2, KL
        lr_z, lr_th = float(self._p['lr_z']), float(self._p['lr_theta'])
        best_loss = np.inf
        self.fit_history = []
        if verbose:
            print('Iterative Updating Starts ...')
        for epoch in range(epochs + 1):
            sample_idx = torch.from_numpy(np.random.choice(n, n, replace=False).astype(np.int32)).to(dev)
            acc.zero_()
            n_steps = 0
            for i in range(0, n_use - b_loc + 1, b_loc):                                            # incomplete last batch skipped (:299)
                idx = sample_idx[i:i + b_loc]
                bg = b_loc * world
                s0 = self._streams(4)
                if world > 1:
                    eng.theta_step(self.data_z, idx, x, y, v, lr_th, seed, s0, apply=False, batch_global=bg, out=out_t)
                    eng.grad_exchange(grad, False)
                    parallel.all_reduce_sum_(grad)
                    eng.grad_exchange(grad, True)
                    eng.theta_apply(lr_th)
                else:
                    eng.theta_step(self.data_z, idx, x, y, v, lr_th, seed, s0, apply=True, out=out_t)
                # NLL gradient of the batch latents with the standard-normal prior (no update), then the joint latent / prior-net step
                eng.z_step(x, y, v, self.data_z, None, None, idx, lr_z, seed, s0 + 1, batch_global=bg, out=out_z, dz_out=dz)
                self._z_t += 1
                self._prior_t += 1
                _lib.check(eng.lib.bgm_bprior_step(eng.h, C.byref(self._prior_cfg), self._norm_mode, self._prior_klw, self._prior_theta.data_ptr(),
                                                   self._prior_m.data_ptr(), self._prior_v.data_ptr(), seg_dev.data_ptr(), self.data_z.data_ptr(),
                                                   idx.data_ptr(), b_loc, bg, parallel.rank() * b_loc, dz.data_ptr(), lr_z, lr_th, self._z_t,
                                                   self._prior_t, seed_shared, s0 + 3, None if world == 1 else pgrad.data_ptr(),
                                                   1 if world == 1 else 0, out_p.data_ptr(), eng._stream()), "bgm_bprior_step")
                if world > 1:
                    parallel.all_reduce_sum_(pgrad)
                    _lib.check(eng.lib.bgm_bprior_apply(eng.h, C.byref(self._prior_cfg), self._prior_klw, self._prior_theta.data_ptr(),
                                                        self._prior_m.data_ptr(), self._prior_v.data_ptr(), pgrad.data_ptr(), lr_th,
                                                        self._prior_t, eng._stream()), "bgm_bprior_apply")
                acc[0] += out_z[0]; acc[1:] += out_p
                n_steps += 1
            a = acc.cpu().numpy() / max(1, n_steps)
            if world > 1:       # a[0] (inv_B = 1 / batch_global) and the prior outputs are this rank's shares of the global batch means
                t_ = torch.tensor([a[0], a[1], a[2]], device=dev, dtype=torch.float64)
                parallel.all_reduce_sum_(t_)
                a[0], a[1], a[2] = float(t_[0]), float(t_[1]), float(t_[2])
            post = float(a[0] - a[2] + a[1] + self._prior_klw * a[3])          # exchange the prior term, add the prior net's KL (:213-215)
            lt = out_t.cpu().numpy()
            self.fit_history.append(dict(epoch=epoch, loss_v=float(lt[0]), loss_mse_v=float(lt[1]), loss_x=float(lt[2]), loss_mse_x=float(lt[3]),
                                         loss_y=float(lt[4]), loss_mse_y=float(lt[5]), loss_postrior_z=post, loss_prior_z=float(a[1]), kl_prior=float(a[3])))
            if verbose:
                print('Epoch [%d/%d]: loss_px_z [%.4f], loss_mse_x [%.4f], loss_py_z [%.4f], loss_mse_y [%.4f], loss_pv_z [%.4f], '
                      'loss_mse_v [%.4f], loss_postrior_z [%.4f]' % (epoch, epochs, lt[2], lt[3], lt[4], lt[5], lt[0], lt[1], post))
            if epoch % epochs_per_eval == 0:
                causal_pre, mse_x, mse_y, mse_v = self._evaluate_dev(x, y, v, self.data_z, n_total, lo_r)
                self.fit_history[-1].update(mse_x=float(mse_x), mse_y=float(mse_y), mse_v=float(mse_v))
                if verbose:
                    print('Epoch [%d/%d]: MSE_x: %.4f, MSE_y: %.4f, MSE_v: %.4f\n' % (epoch, epochs, mse_x, mse_y, mse_v))
                if epoch >= startoff and mse_y < best_loss:
                    best_loss = mse_y
                    self.best_causal_pre = causal_pre
                    self.best_epoch = epoch
                    if self._p['save_model'] and parallel.rank() == 0:
                        self.save_checkpoint(epoch)
                if self._p['save_res'] and parallel.rank() == 0:
                    save_data('{}/causal_pre_at_{}.{}'.format(self.save_dir, epoch, save_format), causal_pre)
        self._pull_weights()

    # ------------------------------------------------------------------ sampling (:497-614)
    def _segments_for(self, n, data_u=None):
        k = int(self.params['n_segments'])
        if data_u is None:
            return np.random.randint(0, k, size=n)                                                # fresh U at predict time (:563-564)
        return np.asarray(data_u).argmax(axis=1)

    def get_log_posterior(self, data_x, data_y, data_v, data_z, data_u, eps=1e-6):
        """log p(z | x, y, v, u) + const, shape (n,) (:497-555): one noisy call of g, h, f and of the prior net on the rows given."""
        seg = torch.from_numpy(self._segments_for(len(data_x), data_u).astype(np.int32)).to(self.engine.device)
        self._set_prior(seg)
        try:
            return CausalBGMBayes.get_log_posterior(self, data_x, data_y, data_v, data_z)
        finally:
            self._clear_prior()

    def metropolis_hastings_sampler(self, data, initial_q_sd=1.0, q_sd=None, burn_in=5000, n_keep=3000, target_acceptance_rate=0.25,
                                    tolerance=0.05, adjustment_interval=50, adaptive_sd=None, window_size=100):
        """(samples [n_keep, n, q], data_u one-hot [n, n_segments]) (:557-614)."""
        segs = self._segments_for(len(data[0]))
        self._set_prior(torch.from_numpy(segs.astype(np.int32)).to(self.engine.device))
        try:
            samples = CausalBGMBayes.metropolis_hastings_sampler(self, data, initial_q_sd=initial_q_sd, q_sd=q_sd, burn_in=burn_in, n_keep=n_keep,
                                                                 target_acceptance_rate=target_acceptance_rate, tolerance=tolerance,
                                                                 adjustment_interval=adjustment_interval, adaptive_sd=adaptive_sd,
                                                                 window_size=window_size)
        finally:
            self._clear_prior()
        return samples, np.eye(int(self.params['n_segments']), dtype=np.float32)[segs]

    def predict(self, data, alpha=0.01, n_mcmc=3000, x_values=None, q_sd=1.0, sample_y=True, bs=100, burn_in=5000, verbose=1):
        """Causal effects with posterior intervals (:348-420): ONE sampler run over all rows with a fresh random U (the panel is one
        block: one perturbation per network call for all rows), effects of every retained draw fused behind it; `bs` only chunked the
        host-side effect pass of the reference and does not change the result."""
        assert 0 < alpha < 1, "The significance level 'alpha' must be greater than 0 and less than 1."
        parallel.check_n_mcmc(n_mcmc)
        binary = bool(self._p['binary_treatment'])
        if not binary and x_values is None:
            raise ValueError("For continuous treatment, 'x_values' must not be None.")
        if x_values is not None:
            x_values = np.array([x_values], dtype=float) if np.isscalar(x_values) else np.array(x_values, dtype=float)
        data_x, data_y, data_v = data
        n = len(data_x)
        eng, dev = self.engine, self.engine.device
        if verbose:
            print('MCMC Latent Variable Sampling ...')
        adaptive = (q_sd is None) or (q_sd <= 0)
        seg_all = parallel.broadcast_(torch.from_numpy(self._segments_for(n).astype(np.int32)).to(dev))      # rank 0's draw is everybody's
        seed = self._next_seed()
        block = max(2, n)
        # Under torch.distributed the rows of the ONE block are sharded like every other class's rows (round 6): the perturbations are
        # keyed by the block, the sign words by the position inside it (mh_run(block_row0 = first local row)), the proposals by the global
        # row -- the result does not depend on the split.  Sessions outside the default-shape sampling kernels keep the replicated run.
        world = parallel.world_size()
        sharded = world > 1 and eng.serves_block_shares()
        lo_r, hi_r = parallel.shard_range(n) if sharded else (0, n)
        x, y, v = self._dev(data_x[lo_r:hi_r]).reshape(-1), self._dev(data_y[lo_r:hi_r]).reshape(-1), self._dev(data_v[lo_r:hi_r])
        n_loc = hi_r - lo_r
        kw = dict(block_row0=lo_r, block_rows_global=n) if sharded else {}
        self._set_prior(seg_all[lo_r:hi_r].contiguous())
        try:
            if binary:
                ite = torch.empty((n_loc, n_mcmc), device=dev, dtype=torch.float32)
                _, acc_tail, tail = self._run_chains(x, y, v, block, burn_in, n_mcmc, q_sd, seed, lo_r, 0, adaptive, effect=2, sample_y=sample_y, ite=ite, **kw)
            else:
                xv = self._dev(x_values.astype(np.float32))
                sums = torch.zeros((len(x_values), n_mcmc), device=dev, dtype=torch.float64)
                _, acc_tail, tail = self._run_chains(x, y, v, block, burn_in, n_mcmc, q_sd, seed, lo_r, 0, adaptive, effect=1, sample_y=sample_y,
                                                     x_values=xv, adrf_sum=sums, **kw)
        finally:
            self._clear_prior()
        self._report_acceptance(float(acc_tail), tail, n, verbose)
        if binary:
            mean, lo, hi = eng.row_mean_quantiles(ite, alpha / 2, 1 - alpha / 2)
            if sharded:
                mean = parallel.all_gather_rows_var(mean.reshape(-1, 1)).reshape(-1)
                lo = parallel.all_gather_rows_var(lo.reshape(-1, 1)).reshape(-1)
                hi = parallel.all_gather_rows_var(hi.reshape(-1, 1)).reshape(-1)
            return mean.cpu().numpy(), torch.stack([lo, hi], dim=1).cpu().numpy()
        if sharded:
            parallel.all_reduce_sum_(sums)
        eff = (sums / float(n)).float().contiguous()
        mean, lo, hi = eng.row_mean_quantiles(eff, alpha / 2, 1 - alpha / 2)
        return mean.cpu().numpy(), torch.stack([lo, hi], dim=1).cpu().numpy()
