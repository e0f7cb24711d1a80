"""BGM -- host-side mirror of the reference class, driving the gfx950 kernels of libbgm_hip.so.

Mirrors /root/reference/src/bayesgm/models/bgm/base.py:
    __init__ :59-121   get_config :123   fit :343   evaluate :445   generate :479
    predict_on_posteriors :511   predict :527   get_log_posterior :666   tfp_mcmc_sampler :709
This class is the deterministic generator (``use_bnn=False``: BaseVariationalNet, networks/base.py:53-117); the EGM warm
start runs on the kernels of csrc/bgm_egm_kernels.h.  ``BGM(params)`` with ``params['use_bnn'] = True``
(BayesianVariationalNet, networks/bnn.py:40-99) returns the subclass models/bgm_bnn.py::BGMBayes (DESIGN_HISTORY.md section 7).
"""
import datetime
import os

import numpy as np
import torch

from .. import parallel
from ..engine import BgmEngine
from ..datasets import Gaussian_sampler

_DEFAULTS = dict(use_bnn=False, g_units=[64] * 5, e_units=[64] * 5, dz_units=[64, 32, 8], dx_units=[64, 32, 8],
                 lr=0.001, lr_theta=0.005, lr_z=0.005, g_d_freq=1, save_model=False, save_res=True, kl_weight=5e-5,
                 use_z_rec=True, alpha=0.0, gamma=0.0)


def _glorot(rs, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rs.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32)


class BGM(object):
    def __new__(cls, params=None, *args, **kwargs):
        if cls is BGM and isinstance(params, dict) and params.get("use_bnn", _DEFAULTS["use_bnn"]):
            from .bgm_bnn import BGMBayes
            return object.__new__(BGMBayes)
        return object.__new__(cls)

    def __init__(self, params, timestamp=None, random_seed=None, device=None):
        self.params = params
        self.timestamp = timestamp
        p = dict(_DEFAULTS)
        p.update(params)
        self._p = p
        random_seed = parallel.shared_seed(random_seed)   # None stays None in a single process; one seed for all ranks otherwise
        if random_seed is not None:
            np.random.seed(random_seed)
        self._rs = np.random.RandomState(random_seed)
        q, xd = int(p["z_dim"]), int(p["x_dim"])
        dims = [q] + list(p["g_units"])
        # BaseVariationalNet parameters: BatchNormalization(z) + Dense stack + mean/var heads, Keras defaults
        self.g = {"bn": {"gamma": np.ones(q, np.float32), "beta": np.zeros(q, np.float32),
                         "mean": np.zeros(q, np.float32), "var": np.ones(q, np.float32)},
                  "trunk": [(_glorot(self._rs, dims[i], dims[i + 1]), np.zeros(dims[i + 1], np.float32))
                            for i in range(len(dims) - 1)],
                  "mean": (_glorot(self._rs, dims[-1], xd), np.zeros(xd, np.float32)),
                  "var": (_glorot(self._rs, dims[-1], xd), np.zeros(xd, np.float32))}
        self.z_sampler = Gaussian_sampler(mean=np.zeros(q), sd=1.0)
        if device is None:     # BGM_DEVICE: dev aid (several ranks on one GPU over gloo)
            device = int(os.environ.get("BGM_DEVICE", os.environ.get("LOCAL_RANK", 0)))
        self.engine = BgmEngine(xd, q, g_units=p["g_units"], device=device)
        from .causalbgm import _disc_norm
        self.engine.set_disc_norm(_disc_norm(p))
        self.engine.set_weights(self.g)
        # params['hmc_precision'] (build option, default "fp32" = the reference's arithmetic): "f16x3" runs the generator's products in
        # predict's log-posterior / HMC kernels in split precision on the fp16 matrix instruction (csrc/bgm_kernels.h)
        if p.get("hmc_precision", "fp32") != "fp32":
            self.engine.set_precision(p["hmc_precision"])
        if self.timestamp is None:
            self.timestamp = datetime.datetime.now().astimezone().strftime('%Y%m%d_%H%M%S')
        self.checkpoint_path = "{}/checkpoints/{}/{}".format(params['output_dir'], params['dataset'], self.timestamp)
        if p['save_model'] and not os.path.exists(self.checkpoint_path):
            os.makedirs(self.checkpoint_path, exist_ok=True)
        self.save_dir = "{}/results/{}/{}".format(params['output_dir'], params['dataset'], self.timestamp)
        if p['save_res'] and not os.path.exists(self.save_dir):
            os.makedirs(self.save_dir, exist_ok=True)
        self.data_z = None
        self._egm_open = False
        self.last_acceptance_rate = None

    def get_config(self):
        return {"params": self.params}

    def _egm_noise_generator(self):
        """Device generator of the EGM reparameterisation noise, keyed by the (rank-shared) seed: the warm start is
        replicated under torch.distributed, so every rank must draw the same noise, and ``BGM(params, random_seed=k)`` must
        be reproducible (the reference seeds tf.random through tf.keras.utils.set_random_seed, bgm/base.py:63-66)."""
        if getattr(self, "_egm_gen", None) is None:
            self._egm_gen = torch.Generator(device=self.engine.device)
            self._egm_gen.manual_seed(int(self._rs.randint(0, 2 ** 31 - 1)))
        return self._egm_gen

    def set_weights(self, g):
        """Install generator parameters (dict with 'bn', 'trunk', 'mean', 'var' as in oracle/nets.init_varnet)."""
        self.g = g
        self.engine.set_weights(g)

    def _dev(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.engine.device, dtype=torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.engine.device)

    # ------------------------------------------------------------------ fit
    def egm_init(self, data, egm_n_iter=10000, batch_size=32, egm_batches_per_eval=500, verbose=1):
        """EGM warm start (bgm/base.py:292-340): g_d_freq LSGAN steps on the discriminators dz_net / dx_net, then one
        step on the generator and the encoder, per iteration -- each step one launch of the kernels of
        csrc/bgm_egm_kernels.h.  The host draws minibatches (Base_sampler), prior samples and interpolation
        coefficients in the reference's order; the reparameterisation noise is drawn on the device.  The session stays
        open afterwards: its encoder serves the Z initialisation e(X) and `evaluate(data_z=None)`."""
        from ..datasets import Base_sampler
        data = np.asarray(data, dtype=np.float32)
        eng = self.engine
        dev = eng.device
        p_ = self._p
        q, xd_ = eng.q, eng.p
        self.data_sampler = Base_sampler(x=data, y=data, v=data, batch_size=batch_size, normalize=False)   # :294
        xd = self._dev(data)

        def mlp(dims):
            return [(_glorot(self._rs, dims[i], dims[i + 1]), np.zeros(dims[i + 1], np.float32)) for i in range(len(dims) - 1)]

        def disc(in_dim, units):                                   # networks/base.py:338-363
            dims = [in_dim] + list(units) + [1]
            return {"W": [_glorot(self._rs, dims[i], dims[i + 1]) for i in range(len(dims) - 1)],
                    "b": [np.zeros(dims[i + 1], np.float32) for i in range(len(dims) - 1)],
                    "gamma": [np.ones(dims[i + 1], np.float32) for i in range(len(dims) - 2)],
                    "beta": [np.zeros(dims[i + 1], np.float32) for i in range(len(dims) - 2)]}
        if self._egm_open:
            eng.egm_end()
            self._egm_open = False
        eng.set_weights(self.g)
        eng.egm_begin(batch_size, p_["e_units"], p_["dz_units"], p_["dx_units"], p_["lr"], p_["gamma"], p_["alpha"],
                      mlp([xd_] + list(p_["e_units"]) + [q]), disc(q, p_["dz_units"]), disc(xd_, p_["dx_units"]))
        self._egm_open = True
        out_d = torch.zeros(3, device=dev)
        out_g = torch.zeros(6, device=dev)
        print('EGM Initialization Starts ...')
        g_d_freq = int(p_['g_d_freq'])
        steps = g_d_freq + 1
        batch_iter = 0
        while batch_iter <= egm_n_iter:
            stop = min(egm_n_iter, (batch_iter // egm_batches_per_eval + 1) * egm_batches_per_eval
                       if batch_iter % egm_batches_per_eval else batch_iter)
            n_it = stop - batch_iter + 1
            x_h = np.empty((n_it, steps, batch_size, xd_), np.float32)
            z_h = np.empty((n_it, steps, batch_size, q), np.float32)
            eps_h = np.empty((n_it, g_d_freq, 2), np.float64)
            for i in range(n_it):                       # host RNG / sampler consumed in the reference's order
                for j in range(steps):
                    x_h[i, j] = self.data_sampler.next_batch()[0]
                    z_h[i, j] = self.z_sampler.get_batch(batch_size)
                    if j < g_d_freq:
                        eps_h[i, j] = np.random.uniform(0.0, 1.0, size=2)
            x_d, z_d = torch.from_numpy(x_h).to(dev), torch.from_numpy(z_h).to(dev)
            noise = torch.randn((n_it, steps + 1, batch_size, xd_), device=dev, generator=self._egm_noise_generator())
            for i in range(n_it):
                for j in range(g_d_freq):
                    eng.egm_disc_step(z_d[i, j], x_d[i, j], noise[i, j], eps_h[i, j, 0], eps_h[i, j, 1], out=out_d)
                eng.egm_gen_step(z_d[i, g_d_freq], x_d[i, g_d_freq], noise[i, g_d_freq], noise[i, g_d_freq + 1], out=out_g)
            batch_iter = stop
            if batch_iter % egm_batches_per_eval == 0:
                if verbose:
                    lg, ld = out_g.cpu().numpy(), out_d.cpu().numpy()
                    print('EGM Initialization Iter [%d] : g_loss_adv[%.4f], e_loss_adv [%.4f], l2_loss_z [%.4f], '
                          'l2_loss_x [%.4f], sd^2_loss[%.4f], g_e_loss [%.4f], dz_loss [%.4f], dx_loss[%.4f], d_loss [%.4f]'
                          % (batch_iter, lg[0], lg[1], lg[2], lg[3], lg[4], lg[5], ld[0], ld[1], ld[2]))
                # evaluation block (:312-317): g_net(e_net(data)) is called with its default training=True, i.e. with the
                # batch statistics of e(data), and moves the BatchNorm moving averages once more
                z_ = eng.egm_encode(xd)
                mu_b, var_b = z_.mean(dim=0).cpu().numpy(), z_.var(dim=0, unbiased=False).cpu().numpy()
                th = eng.egm_read(0)
                eng.egm_sync()
                g_batch = eng.get_weights()
                g_batch["bn"]["mean"], g_batch["bn"]["var"] = mu_b.astype(np.float32), var_b.astype(np.float32)
                th[2 * q:3 * q] = th[2 * q:3 * q] * np.float32(0.99) + mu_b * np.float32(0.01)
                th[3 * q:4 * q] = th[3 * q:4 * q] * np.float32(0.99) + var_b * np.float32(0.01)
                eng.egm_write(0, th)
                # x_rec = mean head of the training-mode call = inference with the batch statistics installed
                eng.set_weights(g_batch)
                x_rec, _ = self._decode(z_, use_x_sd=False)
                print('MSE_x', float(np.mean((data - x_rec) ** 2)))
                eng.egm_sync()
                self.g = eng.get_weights()
                if self._p['save_res']:
                    gen1, var1 = self.generate(nb_samples=5000)          # all ranks: keeps the host RNG in lock step
                    gen12, var12 = self.generate(nb_samples=5000, use_x_sd=False)
                    if parallel.rank() == 0:
                        np.savez('%s/init_data_gen_at_%d.npz' % (self.save_dir, batch_iter), gen1=gen1, gen12=gen12,
                                 z=z_.cpu().numpy(), x_rec=x_rec, var1=var1, var12=var12)
                mse_x = self.evaluate(data=data, use_x_sd=True)
                print('iter [%d/%d]: MSE_x: %.4f\n' % (batch_iter, egm_n_iter, mse_x))
                mse_x = self.evaluate(data=data, use_x_sd=False)
                print('iter [%d/%d]: MSE_x no x_sd: %.4f\n' % (batch_iter, egm_n_iter, mse_x))
                if self._p['save_model'] and parallel.rank() == 0:
                    self.save_checkpoint('egm_init_%d' % batch_iter)
            batch_iter += 1
        eng.egm_sync()
        self.g = eng.get_weights()
        print('EGM Initialization Ends.')

    def fit(self, data, batch_size=32, epochs=100, epochs_per_eval=5, use_egm_init=True, egm_n_iter=20000,
            egm_batches_per_eval=500, verbose=1, host_loop=False):
        """Iterative theta / Z updates (bgm/base.py:343-442).  ``host_loop=False`` (single process): the minibatches of an epoch are
        issued by one library call; ``True`` keeps the per-minibatch calls from Python (same results; the only form under
        torch.distributed, where the all-reduce sits between them).  The incomplete last minibatch of an epoch is
        skipped (:399) and the batch latents take a fresh-slot Adam step (:402, see bgm_fit_kernels.h).

        Under torch.distributed the fit is synchronous data parallel: every rank owns a contiguous
        shard of the rows and their latents, takes `batch_size` of ITS rows per step (global minibatch = world x
        batch_size), the theta gradients -- scaled by the global batch size -- are summed with one all-reduce, and
        every rank applies the same Adam step.  Deviation, stated: the input BatchNorm of each rank normalises with
        the statistics of its local batch; the moving averages are averaged over the ranks when the fit ends.  The EGM
        warm start runs replicated (same seed, same minibatches on every rank)."""
        dist_on = parallel.is_dist()
        world = parallel.world_size()
        if use_egm_init:
            self.egm_init(data, egm_n_iter=egm_n_iter, batch_size=batch_size,
                          egm_batches_per_eval=egm_batches_per_eval, verbose=verbose)
        data = np.asarray(data, dtype=np.float32)
        n_all = len(data)
        lo_r, hi_r = parallel.shard_range(n_all)
        data_loc = data[lo_r:hi_r]
        n = len(data_loc)
        if self._p['save_res'] and parallel.rank() == 0:
            with open('{}/params.txt'.format(self.save_dir), 'w') as f_params:
                f_params.write(str(self.params))
        eng = self.engine
        dev = eng.device
        x = self._dev(data_loc)
        if use_egm_init:
            print('Initialize latent variables Z with e(V)...')
            self.data_z = eng.egm_encode(x).contiguous()                                # :384
        else:
            print('Random initialization of latent variables Z...')
            data_z_init = np.random.normal(0, 1, size=(n_all, eng.q)).astype('float32')     # :388
            self.data_z = self._dev(data_z_init[lo_r:hi_r])
        n_params = eng.fit_begin(n, batch_size)
        if dist_on:
            eng.fit_set_global_batch(batch_size * world)
        n_steps = len(range(0, n_all // world - batch_size + 1, batch_size))    # the same count on every rank
        grad = torch.empty(n_params, device=dev)
        loss = torch.zeros(4, device=dev, dtype=torch.float64)
        self.history_loss = []
        self.fit_history = []       # per-epoch row means of the theta-phase losses (the reference shows them in its progress bar)
        if verbose:
            print('Iterative Updating Starts ...')
        try:
            for epoch in range(epochs + 1):
                sample_idx = torch.from_numpy(np.random.choice(n, n, replace=False).astype(np.int32)).to(dev)
                loss.zero_()
                n_used = 0
                if not dist_on and host_loop is False:      # the minibatch loop inside the library (bgm_bgm_fit_epoch)
                    eng.fit_epoch(x, self.data_z, sample_idx, n_steps, batch_size, self._p['lr_theta'], self._p['lr_z'], loss)
                    n_used = n_steps * batch_size
                for k in (range(n_steps) if n_used == 0 else ()):                   # skip the incomplete last batch
                    idx = sample_idx[k * batch_size:(k + 1) * batch_size]
                    eng.fit_theta_grad(x, self.data_z, idx, grad, loss)
                    if dist_on:
                        parallel.all_reduce_sum_(grad)
                    eng.fit_theta_apply(grad, self._p['lr_theta'])
                    eng.fit_z_step(x, self.data_z, idx, self._p['lr_z'], loss)
                    n_used += batch_size
                l = loss.cpu().numpy() / max(1, n_used)
                self.fit_history.append(dict(epoch=epoch, loss_x=float(l[0] * (world if dist_on else 1)), loss_mse_x=float(l[1] / eng.p),
                                             loss_px_z=float(l[2] * (world if dist_on else 1))))
                if epoch % epochs_per_eval == 0:
                    self.g = eng.get_weights()
                    mse_x = self._evaluate_sharded(data_loc, self.data_z, n_all)
                    self.history_loss.append(mse_x)
                    if verbose and parallel.rank() == 0:
                        print('Epoch [%d/%d]: loss_x [%.4f], loss_mse_x [%.4f], MSE_x: %.4f\n'
                              % (epoch, epochs, l[0] * (world if dist_on else 1), l[1] / eng.p, mse_x))
                    if self._p['save_model'] and parallel.rank() == 0:
                        self.save_checkpoint(epoch)
                    if self._p['save_res'] and parallel.rank() == 0:
                        gen1, var1 = self.generate(nb_samples=5000)
                        gen12, var12 = self.generate(nb_samples=5000, use_x_sd=False)
                        np.savez('%s/data_gen_at_%d.npz' % (self.save_dir, epoch), gen1=gen1, gen12=gen12,
                                 z=self.data_z.cpu().numpy(), var1=var1, var12=var12)
        finally:
            eng.fit_end()
            self.g = eng.get_weights()
            if dist_on:     # identical parameters on every rank; the BatchNorm moving averages saw different local batches
                st = torch.from_numpy(np.concatenate([self.g["bn"]["mean"], self.g["bn"]["var"]]).astype(np.float32)).to(dev)
                parallel.all_reduce_sum_(st)
                st = (st / world).cpu().numpy()
                q = eng.q
                self.g["bn"]["mean"], self.g["bn"]["var"] = st[:q].copy(), st[q:].copy()
                eng.set_weights(self.g)

    def _evaluate_sharded(self, data_loc, data_z_loc, n_all):
        """evaluate() of the rows this rank owns, combined over the ranks (plain evaluate() when not distributed)."""
        z = data_z_loc.cpu().numpy() if isinstance(data_z_loc, torch.Tensor) else np.asarray(data_z_loc, np.float32)
        x_pred, _ = self._decode(z, True)
        sse = torch.tensor([float(np.sum((np.asarray(data_loc, np.float32) - x_pred) ** 2))], dtype=torch.float64,
                           device=self.engine.device)
        parallel.all_reduce_sum_(sse)
        return np.float32(sse.item() / (n_all * data_loc.shape[1]))

    def save_checkpoint(self, epoch):
        """Counterpart of g_net.save_weights(...) (bgm/base.py:431-434): generator parameters as .npz.  The reference's BGM builds a
        tf.train.CheckpointManager (max_to_keep=100, restore-latest at construction, bgm/base.py:108-121) but its fit / egm_init
        never call ckpt_manager.save -- they write these weight files (:335-336, :433) -- so no managed checkpoint ever exists and
        nothing is auto-restored; the plain files here are that behaviour, not an omission."""
        path = os.path.join(self.checkpoint_path, "weights_at_%s_generator.npz" % epoch)
        flat = {"bn_" + k: v for k, v in self.g["bn"].items()}
        for i, (W, b) in enumerate(self.g["trunk"]):
            flat["trunk_W%d" % i], flat["trunk_b%d" % i] = W, b
        for k in ("mean", "var"):
            flat[k + "_W"], flat[k + "_b"] = self.g[k]
        np.savez(path, **flat)
        print('Saving checkpoint for epoch {} at {}'.format(epoch, path))
        return path

    def load_checkpoint(self, path):
        """Install the generator parameters written by `save_checkpoint` (the reference restores with
        g_net.load_weights / tf.train.Checkpoint.restore)."""
        d = np.load(path)
        n_trunk = len(self.g["trunk"])
        g = {"bn": {k: np.asarray(d["bn_" + k], np.float32) for k in ("gamma", "beta", "mean", "var")},
             "trunk": [(np.asarray(d["trunk_W%d" % i], np.float32), np.asarray(d["trunk_b%d" % i], np.float32))
                       for i in range(n_trunk)],
             "mean": (np.asarray(d["mean_W"], np.float32), np.asarray(d["mean_b"], np.float32)),
             "var": (np.asarray(d["var_W"], np.float32), np.asarray(d["var_b"], np.float32))}
        self.set_weights(g)

    # ------------------------------------------------------------------ inference helpers
    def get_log_posterior(self, data_z, data_x, ind_x1=None, obs_mask=None):
        """log p(z | x_obs) + const (bgm/base.py:665-705).  Missing cells: NaN in data_x, or the reference's
        (ind_x1 [n,K], obs_mask [n,K]) index form, which is converted to the NaN form."""
        x = np.array(data_x, dtype=np.float32, copy=True)
        if ind_x1 is not None:
            ind = np.asarray(ind_x1)
            if ind.ndim == 1:
                ind = np.broadcast_to(ind[None, :], (x.shape[0], ind.shape[0]))
            keep = np.zeros(x.shape, bool)
            mk = np.ones(ind.shape, bool) if obs_mask is None else (np.asarray(obs_mask) > 0)
            rows = np.repeat(np.arange(x.shape[0])[:, None], ind.shape[1], 1)
            keep[rows[mk], ind[mk]] = True
            x[~keep] = np.nan
        return self.engine.logpost(self._dev(data_z), self._dev(x)).cpu().numpy()

    def tfp_mcmc_sampler(self, data, ind_x1=None, n_mcmc=3000, burn_in=5000, step_size=0.01, num_leapfrog_steps=10,
                         seed=42):
        """Posterior samples of Z, shape (n_mcmc, n, z_dim) (bgm/base.py:709-830)."""
        x = np.array(data, dtype=np.float32, copy=True)
        if ind_x1 is not None:
            keep = np.zeros(x.shape, bool)
            if len(ind_x1) > 0 and isinstance(ind_x1[0], (list, tuple, np.ndarray)):
                assert len(ind_x1) == x.shape[0], f"len(ind_x1)={len(ind_x1)} != n_samples={x.shape[0]}"
                assert max(len(r) for r in ind_x1) > 0, "No observed features"
                for i, r in enumerate(ind_x1):
                    keep[i, list(r)] = True
            else:
                keep[:, list(ind_x1)] = True
            x[~keep] = np.nan
        out = self.engine.hmc_sample(self._dev(x), n_mcmc, burn_in, step_size, num_leapfrog_steps, seed)
        self.last_acceptance_rate = float(out["acc_count"][burn_in:].sum().item()) / max(1, n_mcmc * x.shape[0])
        print(f"TFP MCMC Acceptance Rate: {self.last_acceptance_rate:.4f}")
        return out["draws"].cpu().numpy()

    def predict_on_posteriors(self, data_posterior_z, seed=0):
        """x ~ N(mu(z), sigma^2(z)) for every draw (bgm/base.py:511-525) -> (n_mcmc, n, x_dim)."""
        _, full = self.engine.predict_draws(self._dev(data_posterior_z), 0, seed, want_full=True)
        return full.cpu().numpy()

    def generate(self, nb_samples=1000, use_x_sd=True):
        """(data_x_gen, sigma_square_x) from z ~ N(0, I) (bgm/base.py:478-509)."""
        z = np.random.normal(0, 1, size=(1, nb_samples, self.engine.q)).astype(np.float32)
        return self._decode(z[0], use_x_sd)

    def _decode(self, z, use_x_sd, seed=None):
        """(x, sigma^2) for latent rows z with g_net(training=False) on the device (bgm/base.py:468-473,503-508)."""
        if seed is None:      # the mean path needs no noise: leave NumPy's global stream alone
            seed = int(np.random.randint(0, 2 ** 31 - 1)) if use_x_sd else 0
        zt = self._dev(z)[None]
        _, full, var = self.engine.predict_draws(zt, 0, seed, want_full=True, want_var=True, add_noise=use_x_sd)
        return full[0].cpu().numpy(), var[0].cpu().numpy()

    def evaluate(self, data, data_z=None, use_x_sd=True):
        """mse_x between data and its reconstruction (bgm/base.py:444-476)."""
        if data_z is None:
            if not self._egm_open:
                raise RuntimeError("BGM.evaluate(data_z=None) needs the encoder trained by egm_init(); pass data_z")
            data_z = self.engine.egm_encode(self._dev(data))
        z = data_z.cpu().numpy() if isinstance(data_z, torch.Tensor) else np.asarray(data_z, np.float32)
        x_pred, _ = self._decode(z, use_x_sd)
        return np.float32(np.mean((np.asarray(data, np.float32) - x_pred) ** 2))

    # ------------------------------------------------------------------ predict
    def predict(self, data, alpha=0.05, return_samples=False, bs=100, n_mcmc=5000, burn_in=5000, step_size=0.01,
                num_leapfrog_steps=10, seed=42, max_draw_bytes=64 << 30):
        """Posterior-predictive imputation of the NaN cells (bgm/base.py:527-663).

        HMC burn-in (with the shared step-size adaptation) runs over ALL rows at once as in the reference;
        the sampling phase then runs in row blocks sized so that the latent draws and the predictive cells of
        a block stay below ``max_draw_bytes`` (the reference materialises [n_mcmc, n, x_dim] on the host)."""
        assert 0 < alpha < 1, "The significance level 'alpha' must be greater than 0 and less than 1."
        parallel.check_n_mcmc(n_mcmc)
        import time as _time
        _t = {"_last": _time.perf_counter()}

        def _mark(name):   # wall-clock phase breakdown (diagnostics only; kept in self.last_predict_timing)
            torch.cuda.synchronize(self.engine.device)
            now = _time.perf_counter()
            _t[name] = _t.get(name, 0.0) + now - _t["_last"]
            _t["_last"] = now
        data_np = data.cpu().numpy() if isinstance(data, torch.Tensor) else np.asarray(data, dtype=np.float32)
        data_np = data_np.astype(np.float32)
        n, p = data_np.shape
        eng = self.engine
        dev = eng.device
        miss = np.isnan(data_np)
        lo_r, hi_r = parallel.shard_range(n)
        x = self._dev(data_np[lo_r:hi_r])
        n_loc = hi_r - lo_r
        q = eng.q
        # ---- burn-in over all local rows, step size shared by all chains of all ranks
        state = torch.empty((n_loc, q), device=dev)
        logp = torch.empty(n_loc, device=dev)
        grad = torch.empty((n_loc, q), device=dev)
        step = torch.full((1,), float(step_size), device=dev)
        total = burn_in + n_mcmc
        acc_prob = torch.zeros(total, device=dev, dtype=torch.float64)
        acc_count = torch.zeros(total, device=dev, dtype=torch.int32)
        n_adapt = int(burn_in * 0.8)
        _mark("setup_h2d")
        for it in range(n_adapt):
            eng.hmc_run(x, state, logp, grad, step, it, 1, burn_in, num_leapfrog_steps, seed, init=(it == 0),
                        row_base=lo_r, acc_prob=acc_prob, acc_count=acc_count)
            parallel.all_reduce_sum_(acc_prob[it:it + 1])
            eng.hmc_adapt(step, acc_prob, it, n)
        if burn_in > n_adapt:
            eng.hmc_run(x, state, logp, grad, step, n_adapt, burn_in - n_adapt, burn_in, num_leapfrog_steps, seed,
                        init=(n_adapt == 0), row_base=lo_r, acc_prob=acc_prob, acc_count=acc_count)
        _mark("burn_in")
        # ---- sampling + predictive draws per row block (slot maps, moments and the imputation stay in HBM)
        miss_dev = torch.isnan(x)
        k_row_dev = miss_dev.sum(dim=1)
        k_max = k_row_dev.max().reshape(1) if n_loc else torch.zeros(1, dtype=torch.int64, device=dev)
        k_slots = int(parallel.all_reduce_max_(k_max).item())   # same slot width on every rank (rows are gathered later)
        slot_dev = (torch.cumsum(miss_dev, dim=1, dtype=torch.int32) - 1).to(torch.int32)   # k-th missing cell -> slot k
        slot_dev.masked_fill_(~miss_dev, -1)
        per_row = 4 * n_mcmc * (q + max(k_slots, 1) + (p if return_samples else 0))
        rows_blk = max(16, int(max_draw_bytes // max(1, per_row)))
        quantum = 16 * 24 * torch.cuda.get_device_properties(dev).multi_processor_count   # whole launch waves (8 / 12 waves per CU)
        if rows_blk > quantum:
            rows_blk -= rows_blk % quantum
        means = torch.zeros((n_loc, max(k_slots, 1)), device=dev)
        los = torch.zeros_like(means)
        his = torch.zeros_like(means)
        samples = []
        _mark("slots")
        for s in range(0, n_loc, rows_blk):
            e = min(s + rows_blk, n_loc)
            draws = torch.empty((n_mcmc, e - s, q), device=dev)
            eng.hmc_run(x[s:e], state[s:e], logp[s:e], grad[s:e], step, burn_in, n_mcmc, burn_in, num_leapfrog_steps,
                        seed, init=(burn_in == 0), row_base=lo_r + s, acc_count=acc_count, draws=draws)
            cells = full = None
            if k_slots > 0 or return_samples:
                cells, full = eng.predict_draws(draws, burn_in, seed, slot=slot_dev[s:e] if k_slots > 0 else None,
                                                k_slots=k_slots, want_full=return_samples, row_base=lo_r + s)
            if k_slots > 0:
                mean, lo, hi = eng.row_mean_quantiles(cells, alpha / 2.0, 1.0 - alpha / 2.0)
                means[s:e] = mean.reshape(e - s, k_slots)
                los[s:e] = lo.reshape(e - s, k_slots)
                his[s:e] = hi.reshape(e - s, k_slots)
            if return_samples:
                samples.append(full.cpu().numpy())
            del draws, cells, full
        _mark("sample_predict_quantiles")
        acc = acc_count[burn_in:].sum().double().reshape(1)
        parallel.all_reduce_sum_(acc)
        self.last_acceptance_rate = float(acc.item()) / max(1, n_mcmc * n)
        print(f"TFP MCMC Acceptance Rate: {self.last_acceptance_rate:.4f}")
        # imputation: observed cells as given, missing cells = posterior-predictive mean
        imputed_dev = torch.where(miss_dev, means.gather(1, slot_dev.clamp(min=0).long()), torch.nan_to_num(x, nan=0.0)) \
            if k_slots > 0 else x.clone()
        if parallel.is_dist():
            means, los, his = (parallel.all_gather_rows(a_, n) for a_ in (means, los, his))
            imputed_dev = parallel.all_gather_rows(imputed_dev, n)
        los, his = los.cpu().numpy(), his.cpu().numpy()
        # ---- assemble the reference's return values
        same_pattern = bool(np.all(miss == miss[0]))
        if same_pattern:
            mi = np.where(miss[0])[0]
            if mi.size == 0:
                pred_interval = np.zeros((n, 0, 2), dtype=np.float32)
            else:
                pred_interval = np.stack([los[:, :mi.size], his[:, :mi.size]], axis=-1)
        else:
            k_row = miss.sum(axis=1)
            used = np.arange(los.shape[1])[None, :] < k_row[:, None]
            flat = np.stack([los[used], his[used]], axis=-1).astype(np.float32)          # row-major: row i's cells together
            pred_interval = np.split(flat, np.cumsum(k_row)[:-1]) if n else []
        _mark("intervals")
        self.last_predict_timing = {k: v for k, v in _t.items() if k != "_last"}
        if return_samples:
            full = np.concatenate(samples, axis=1)
            if parallel.is_dist():
                full = parallel.all_gather_rows(torch.from_numpy(np.ascontiguousarray(full.transpose(1, 0, 2))).to(dev),
                                                n).cpu().numpy().transpose(1, 0, 2)
            return full, pred_interval
        imputed = imputed_dev.cpu().numpy()
        _mark("impute_assembly")
        self.last_predict_timing = {k: v for k, v in _t.items() if k != "_last"}
        return imputed, pred_interval
