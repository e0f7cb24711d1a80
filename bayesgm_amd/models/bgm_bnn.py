"""BGM with the Bayesian generator (``params['use_bnn'] = True``) -- host-side mirror of the reference class.

Mirrors /root/reference/src/bayesgm/models/bgm/base.py with g_net = BayesianVariationalNet (networks/bnn.py:40-99; :67-69 of
base.py): fit :343 (update_g_net :145 with the KL term :155-157, update_latent_variable_sgd :167), egm_init :292, evaluate :445,
generate :479, predict_on_posteriors :511, predict :527, get_log_posterior :666, tfp_mcmc_sampler :709.  e_net, dz_net and dx_net
stay deterministic.  Kernels: csrc/bgmb_kernels.h, csrc/bgmb_egm_kernels.h through the C ABI bgm_bvn_*; arithmetic and the
counter-based Flipout noise are restated in oracle/bgm_bnn.py.

DenseFlipout perturbs the kernels in EVERY call, also with training=False (tfp.layers.DenseFlipout._apply_variational_kernel draws
the perturbation and the sign vectors inside `call`, with stateful random ops when the layer's seed is None, and has no training
switch), so evaluate / generate / the predictive draws are stochastic in the weights, as in the reference.

The HMC target (``params['bnn_mcmc_noise']``; DESIGN_HISTORY.md section 7b gives the argument in full).  Under tfp.mcmc.sample_chain the
target_log_prob_fn is traced once and EXECUTED at every leapfrog step, so as written (bgm/base.py:709-830) every gradient
evaluation of a transition sees a different weight draw while the cached log-prob of the current state stays: the acceptance ratio
compares two different perturbations, the acceptance probability is bounded away from 1 however small the step (~0.54 measured),
SimpleStepSizeAdaptation (target 0.75) shrinks the step by 1/1.01 at each of its 0.8 burn_in steps and the chains stop moving (draw
variance 5e-11; imputations worse than the column mean).  tfp.mcmc.HamiltonianMonteCarlo documents target_log_prob_fn as a function
of the state that returns its log-density -- a deterministic function; a fresh draw per call is outside that contract and outside
anything Metropolis-Hastings corrects for -- and what predict() documents itself to return (posterior draws of Z given the observed
entries) exists only for a fixed generator.  So, decided the way ``bnn_norm`` was:
  'frozen' (DEFAULT)  one weight draw per HMC run (generator call 0 for every gradient evaluation): a deterministic target, the
                      reading on which the documented behaviour holds (adaptation reaches 0.75, imputation beats the column mean);
  'fresh'             the reference as executed, an explicit choice: reproduced exactly, with a warning that the chains freeze.
Noise keys: the fit uses random_seed (stream 2t for the
theta step and 2t + 1 for the Z step of minibatch t), the EGM warm start random_seed + 2^40, predict the `seed` argument.
"""
import datetime
import os

import numpy as np
import torch

from .. import parallel
from ..bvn_engine import BvnEngine, STREAM_PREDICT, STREAM_DECODE, flatten_vnet
from ..datasets import Gaussian_sampler
from .bgm import BGM, _DEFAULTS, _glorot


def block_plan(lo_r, n_loc, bs, rows_chunk):
    """Row chunks of the sampling phase and the predictive calls inside them for the local rows [lo_r, lo_r + n_loc) of a
    panel whose predictive calls are the blocks of `bs` GLOBAL rows: [(s, e, [(b, be, block, off), ...]), ...] with local
    row ranges [s, e) / [b, be); `off` = position of global row lo_r + b inside its block.  Chunks end on block
    boundaries, so no call spans two chunks; a block split over two ranks is two calls with the same stream (block id)
    whose Flipout signs are keyed by the position inside the block."""
    plan = []
    s = 0
    rows_chunk = max(bs, rows_chunk // bs * bs)
    while s < n_loc:
        g0 = lo_r + s
        e = min(n_loc, s + rows_chunk - (g0 % bs))
        calls = []
        b = s
        while b < e:
            gb = lo_r + b
            blk, off = gb // bs, gb % bs
            be = min(e, b + bs - off)
            calls.append((b, be, blk, off))
            b = be
        plan.append((s, e, calls))
        s = e
    return plan


class BGMBayes(BGM):
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
        self._seed = int(random_seed or 0) & 0xFFFFFFFF
        q, xd = int(p["z_dim"]), int(p["x_dim"])
        dims = [q] + list(p["g_units"])

        def flip(i, o):      # tfp default_mean_field_normal_fn: loc ~ N(0, 0.1^2), rho ~ N(-3, 0.1^2); bias loc ~ N(0, 0.1^2)
            return ((0.1 * self._rs.standard_normal((i, o))).astype(np.float32),
                    (-3.0 + 0.1 * self._rs.standard_normal((i, o))).astype(np.float32),
                    (0.1 * self._rs.standard_normal(o)).astype(np.float32))
        self.g = {"gamma": np.ones(q, np.float32), "beta": np.zeros(q, np.float32), "mean_mv": np.zeros(q, np.float32),
                  "var_mv": np.ones(q, np.float32), "trunk": [flip(dims[i], dims[i + 1]) for i in range(len(dims) - 1)],
                  "mean": flip(dims[-1], xd), "var": flip(dims[-1], xd)}
        self.z_sampler = Gaussian_sampler(mean=np.zeros(q), sd=1.0)
        if device is None:
            device = int(os.environ.get("BGM_DEVICE", os.environ.get("LOCAL_RANK", 0)))
        self._max_batch = int(params.get("max_batch", 256))      # rows of a minibatch per rank the session is sized for (up to 4096; bgm/base.py:343 takes any batch_size)
        mode = params.get("bnn_mcmc_noise", "frozen")
        if "bnn_mcmc_noise" not in params:
            from .. import diagnostics
            diagnostics.notice_once("bnn_mcmc_noise", "params['bnn_mcmc_noise'] not given: HMC runs on ONE weight perturbation per run ('frozen'); "
                                    "'fresh' re-perturbs at every gradient evaluation as the reference is written (DESIGN_HISTORY.md section 7b)")
        if mode not in ("fresh", "frozen"):
            raise ValueError("params['bnn_mcmc_noise'] must be 'fresh' or 'frozen'")
        self._mcmc_noise = mode
        self._warned_noise = False
        self.engine = BvnEngine(xd, q, g_units=p["g_units"], kl_weight=p["kl_weight"], max_batch=self._max_batch, device=device,
                                hmc_frozen_noise=(mode == "frozen"))
        from .causalbgm import _disc_norm
        self.engine.set_disc_norm(_disc_norm(p))
        self.engine.begin(self.g)
        # params['hmc_precision'] (build option, default "fp32" = the reference's arithmetic): "f16x3" runs the generator's products of
        # the frozen-noise HMC in split fp16 (bgmfx_kernels.h); raises where that kernel does not serve the model
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
        self._egm_steps = 0
        self._fit_t = 0
        self.last_acceptance_rate = None

    def set_weights(self, g):
        """Install generator parameters (dict as oracle/bgm_bnn.init_vnet); Adam slots restart."""
        self.g = g
        if self._egm_open:
            self.engine.egm_end()
            self._egm_open = False
        self.engine.begin(g)

    def _sync_g(self):
        self.g = self.engine.get_net()
        return self.g

    # ------------------------------------------------------------------ EGM warm start
    def egm_init(self, data, egm_n_iter=10000, batch_size=32, egm_batches_per_eval=500, verbose=1):
        """EGM warm start (bgm/base.py:292-340): g_d_freq discriminator steps, then one generator / encoder step per
        iteration, each one launch (csrc/bgmb_egm_kernels.h).  Host RNG consumption as in the deterministic class."""
        from ..datasets import Base_sampler
        data = np.asarray(data, dtype=np.float32)
        eng = self.engine
        dev = eng.device
        p_ = self._p
        q, xd_ = eng.q, eng.p
        if batch_size > self._max_batch:
            raise ValueError("bayesgm_amd: this session is sized for minibatches of at most %d rows; create the model with params['max_batch'] >= %d (up to 4096)" % (self._max_batch, batch_size))
        self.data_sampler = Base_sampler(x=data, y=data, v=data, batch_size=batch_size, normalize=False)
        xd = self._dev(data)

        def mlp(dims):
            return [(_glorot(self._rs, dims[i], dims[i + 1]), np.zeros(dims[i + 1], np.float32)) for i in range(len(dims) - 1)]

        def disc(in_dim, units):
            dims = [in_dim] + list(units) + [1]
            return {"W": [_glorot(self._rs, dims[i], dims[i + 1]) for i in range(len(dims) - 1)],
                    "b": [np.zeros(dims[i + 1], np.float32) for i in range(len(dims) - 1)],
                    "gamma": [np.ones(dims[i + 1], np.float32) for i in range(len(dims) - 2)],
                    "beta": [np.zeros(dims[i + 1], np.float32) for i in range(len(dims) - 2)]}
        if self._egm_open:
            eng.egm_end()
            self._egm_open = False
        eng.egm_begin(batch_size, p_["e_units"], p_["dz_units"], p_["dx_units"], p_["lr"], p_["gamma"], p_["alpha"],
                      mlp([xd_] + list(p_["e_units"]) + [q]), disc(q, p_["dz_units"]), disc(xd_, p_["dx_units"]))
        self._egm_open = True
        key = self._seed + (1 << 40)
        out_d = torch.zeros(3, device=dev)
        out_g = torch.zeros(6, device=dev)
        print('EGM Initialization Starts ...')
        g_d_freq = int(p_['g_d_freq'])
        steps = g_d_freq + 1
        batch_iter = 0
        s = self._egm_steps
        while batch_iter <= egm_n_iter:
            stop = min(egm_n_iter, (batch_iter // egm_batches_per_eval + 1) * egm_batches_per_eval
                       if batch_iter % egm_batches_per_eval else batch_iter)
            n_it = stop - batch_iter + 1
            x_h = np.empty((n_it, steps, batch_size, xd_), np.float32)
            z_h = np.empty((n_it, steps, batch_size, q), np.float32)
            eps_h = np.empty((n_it, g_d_freq, 2), np.float64)
            for i in range(n_it):
                for j in range(steps):
                    x_h[i, j] = self.data_sampler.next_batch()[0]
                    z_h[i, j] = self.z_sampler.get_batch(batch_size)
                    if j < g_d_freq:
                        eps_h[i, j] = np.random.uniform(0.0, 1.0, size=2)
            x_d, z_d = torch.from_numpy(x_h).to(dev), torch.from_numpy(z_h).to(dev)
            noise = torch.randn((n_it, steps + 1, batch_size, xd_), device=dev, generator=self._egm_noise_generator())
            for i in range(n_it):
                for j in range(g_d_freq):
                    eng.egm_disc_step(z_d[i, j], x_d[i, j], noise[i, j], eps_h[i, j, 0], eps_h[i, j, 1], key, 2 * s, out=out_d)
                    s += 1
                eng.egm_gen_step(z_d[i, g_d_freq], x_d[i, g_d_freq], noise[i, g_d_freq], noise[i, g_d_freq + 1], key, 2 * s, out=out_g)
                s += 1
            batch_iter = stop
            if batch_iter % egm_batches_per_eval == 0:
                if verbose:
                    lg, ld = out_g.cpu().numpy(), out_d.cpu().numpy()
                    print('EGM Initialization Iter [%d] : g_loss_adv[%.4f], e_loss_adv [%.4f], l2_loss_z [%.4f], '
                          'l2_loss_x [%.4f], sd^2_loss[%.4f], g_e_loss [%.4f], dz_loss [%.4f], dx_loss[%.4f], d_loss [%.4f]'
                          % (batch_iter, lg[0], lg[1], lg[2], lg[3], lg[4], lg[5], ld[0], ld[1], ld[2]))
                # evaluation block (:312-317): g_net(e_net(data)) runs with its default training=True -- batch statistics of
                # e(data), one more move of the moving averages -- emulated by installing the batch statistics for one
                # inference-mode call
                z_ = eng.egm_encode(xd)
                mu_b, var_b = z_.mean(dim=0).cpu().numpy(), z_.var(dim=0, unbiased=False).cpu().numpy()
                th = eng.egm_read(0)
                n_g = eng.n_params
                tmp = th[:n_g].copy()
                tmp[2 * q:3 * q], tmp[3 * q:4 * q] = mu_b, var_b
                eng.write(tmp)
                x_rec, _ = self._decode(z_, use_x_sd=False)
                print('MSE_x', float(np.mean((data - x_rec) ** 2)))
                th[2 * q:3 * q] = th[2 * q:3 * q] * np.float32(0.99) + mu_b * np.float32(0.01)
                th[3 * q:4 * q] = th[3 * q:4 * q] * np.float32(0.99) + var_b * np.float32(0.01)
                eng.egm_write(0, th)
                eng.egm_sync()
                self._sync_g()
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
        self._egm_steps = s
        eng.egm_sync()
        self._sync_g()
        print('EGM Initialization Ends.')

    # ------------------------------------------------------------------ fit
    def fit(self, data, batch_size=32, epochs=100, epochs_per_eval=5, use_egm_init=True, egm_n_iter=20000,
            egm_batches_per_eval=500, verbose=1):
        """Iterative theta / Z updates (bgm/base.py:343-442) with the Bayesian generator.  Data parallel as the deterministic
        class: local minibatches, one all-reduce of the generator gradient per step, the moving BatchNorm statistics averaged
        over the ranks at the end; the Flipout noise of a step is the same on every rank (per-row signs by batch position)."""
        dist_on = parallel.is_dist()
        world = parallel.world_size()
        if batch_size > self._max_batch:
            raise ValueError("bayesgm_amd: this session is sized for minibatches of at most %d rows; create the model with params['max_batch'] >= %d (up to 4096)" % (self._max_batch, batch_size))
        if use_egm_init:
            self.egm_init(data, egm_n_iter=egm_n_iter, batch_size=batch_size, egm_batches_per_eval=egm_batches_per_eval,
                          verbose=verbose)
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
            self.data_z = eng.egm_encode(x).contiguous()
        else:
            print('Random initialization of latent variables Z...')
            data_z_init = np.random.normal(0, 1, size=(n_all, eng.q)).astype('float32')
            self.data_z = self._dev(data_z_init[lo_r:hi_r])
        n_steps = len(range(0, n_all // world - batch_size + 1, batch_size))
        gbuf = torch.empty(eng.n_params, device=dev) if dist_on else None
        out_t, out_z = torch.zeros(2, device=dev), torch.zeros(1, device=dev)
        loss = torch.zeros(3, device=dev, dtype=torch.float64)
        self.history_loss = []
        key = self._seed
        if verbose:
            print('Iterative Updating Starts ...')
        for epoch in range(epochs + 1):
            sample_idx = torch.from_numpy(np.random.choice(n, n, replace=False).astype(np.int32)).to(dev)
            loss.zero_()
            for k in range(n_steps):
                idx = sample_idx[k * batch_size:(k + 1) * batch_size]
                t = self._fit_t
                if dist_on:
                    eng.theta_step(x, self.data_z, idx, self._p['lr_theta'], key, 2 * t, batch_global=batch_size * world,
                                   apply=False, out=out_t)
                    eng.grad_exchange(gbuf, False)
                    parallel.all_reduce_sum_(gbuf)
                    eng.grad_exchange(gbuf, True)
                    eng.theta_apply(self._p['lr_theta'])
                else:
                    eng.theta_step(x, self.data_z, idx, self._p['lr_theta'], key, 2 * t, out=out_t)
                eng.z_step(x, self.data_z, idx, self._p['lr_z'], key, 2 * t + 1, batch_global=batch_size * world, out=out_z)
                loss[:2] += out_t.double()
                loss[2:] += out_z.double()
                self._fit_t += 1
            if epoch % epochs_per_eval == 0:
                self._sync_g()
                mse_x = self._evaluate_sharded(data_loc, self.data_z, n_all)
                self.history_loss.append(mse_x)
                if verbose and parallel.rank() == 0:
                    l = loss.cpu().numpy() / max(1, n_steps)
                    print('Epoch [%d/%d]: loss_x [%.4f], loss_mse_x [%.4f], loss_postrior_z [%.4f], MSE_x: %.4f\n'
                          % (epoch, epochs, l[0], l[1], l[2], mse_x))
                if self._p['save_model'] and parallel.rank() == 0:
                    self.save_checkpoint(epoch)
                if self._p['save_res'] and parallel.rank() == 0:
                    gen1, var1 = self.generate(nb_samples=5000)
                    gen12, var12 = self.generate(nb_samples=5000, use_x_sd=False)
                    np.savez('%s/data_gen_at_%d.npz' % (self.save_dir, epoch), gen1=gen1, gen12=gen12,
                             z=self.data_z.cpu().numpy(), var1=var1, var12=var12)
        if dist_on:      # identical parameters on every rank; the moving statistics saw different local batches
            th = torch.from_numpy(eng.read(0)).to(dev)
            q = eng.q
            st = th[2 * q:4 * q].clone()
            parallel.all_reduce_sum_(st)
            th[2 * q:4 * q] = st / world
            eng.write(th.cpu().numpy())
        self._sync_g()

    def save_checkpoint(self, epoch):
        """Counterpart of g_net.save_weights(...) (bgm/base.py:431-434): generator parameters as .npz.  The reference's BGM builds a
        tf.train.CheckpointManager (max_to_keep=100, restore-latest at construction, bgm/base.py:108-121) but its fit / egm_init
        never call ckpt_manager.save -- they write these weight files (:335-336, :433) -- so no managed checkpoint ever exists and
        nothing is auto-restored; the plain files here are that behaviour, not an omission."""
        path = os.path.join(self.checkpoint_path, "weights_at_%s_generator.npz" % epoch)
        g = self._sync_g()
        flat = {k: g[k] for k in ("gamma", "beta", "mean_mv", "var_mv")}
        for i, L in enumerate(list(g["trunk"]) + [g["mean"], g["var"]]):
            name = "trunk%d" % i if i < len(g["trunk"]) else ("mean" if i == len(g["trunk"]) else "var")
            flat[name + "_loc"], flat[name + "_rho"], flat[name + "_bias"] = L
        np.savez(path, **flat)
        print('Saving checkpoint for epoch {} at {}'.format(epoch, path))
        return path

    def load_checkpoint(self, path):
        d = np.load(path)
        g = {k: d[k] for k in ("gamma", "beta", "mean_mv", "var_mv")}
        T = len(self._p["g_units"])
        g["trunk"] = [(d["trunk%d_loc" % i], d["trunk%d_rho" % i], d["trunk%d_bias" % i]) for i in range(T)]
        for k in ("mean", "var"):
            g[k] = (d[k + "_loc"], d[k + "_rho"], d[k + "_bias"])
        self.set_weights(g)

    # ------------------------------------------------------------------ inference helpers
    def _warn_fresh_noise(self):
        if self._mcmc_noise == "fresh" and not self._warned_noise:
            import warnings
            warnings.warn("bayesgm_amd: params['bnn_mcmc_noise'] = 'fresh' re-perturbs the generator at every HMC gradient evaluation, as "
                          "the reference executes it: the acceptance probability stays below the adaptation target whatever the step, "
                          "SimpleStepSizeAdaptation shrinks the step geometrically and the chains freeze (DESIGN_HISTORY.md section 7b); "
                          "the default 'frozen' samples on one weight draw per HMC run instead.")
            self._warned_noise = True

    def _new_seed(self):
        return int(np.random.randint(0, 2 ** 31 - 1))

    def get_log_posterior(self, data_z, data_x, ind_x1=None, obs_mask=None, seed=None):
        """log p(z | x_obs) + const for ONE perturbed generator call (bgm/base.py:665-705)."""
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
        return self.engine.logpost(self._dev(data_z), self._dev(x), self._new_seed() if seed is None else seed, 0).cpu().numpy()

    def tfp_mcmc_sampler(self, data, ind_x1=None, n_mcmc=3000, burn_in=5000, step_size=0.01, num_leapfrog_steps=10, seed=42):
        """Posterior samples of Z, shape (n_mcmc, n, z_dim) (bgm/base.py:709-830), stochastic target."""
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
        self._warn_fresh_noise()
        out = self.engine.hmc_sample(self._dev(x), n_mcmc, burn_in, step_size, num_leapfrog_steps, seed)
        self.last_acceptance_rate = float(out["acc_count"][burn_in:].sum().item()) / max(1, n_mcmc * x.shape[0])
        print(f"TFP MCMC Acceptance Rate: {self.last_acceptance_rate:.4f}")
        return out["draws"].cpu().numpy()

    def predict_on_posteriors(self, data_posterior_z, seed=0):
        """x ~ N(mu(z), sigma^2(z)) for every draw, ONE perturbed generator call (bgm/base.py:511-525)."""
        _, full = self.engine.decode(self._dev(data_posterior_z), seed, STREAM_PREDICT, want_full=True)
        return full.cpu().numpy()

    def _decode(self, z, use_x_sd, seed=None):
        """(x, sigma^2) for latent rows z with one g_net(training=False) call (bgm/base.py:468-473, 503-508)."""
        if seed is None:
            seed = self._new_seed()
        zt = self._dev(z)[None]
        _, full, var = self.engine.decode(zt, seed, STREAM_DECODE, want_full=True, want_var=True, add_noise=use_x_sd)
        return full[0].cpu().numpy(), var[0].cpu().numpy()

    # ------------------------------------------------------------------ predict
    def predict(self, data, alpha=0.05, return_samples=False, bs=100, n_mcmc=5000, burn_in=5000, step_size=0.01,
                num_leapfrog_steps=10, seed=42, max_draw_bytes=16 << 30):
        """Posterior-predictive imputation of the NaN cells (bgm/base.py:527-663).  HMC over ALL rows as in the reference
        (one generator call per gradient evaluation), then one predictive generator call per block of `bs` rows."""
        assert 0 < alpha < 1, "The significance level 'alpha' must be greater than 0 and less than 1."
        parallel.check_n_mcmc(n_mcmc)
        self._warn_fresh_noise()
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
        bs = max(1, int(bs))
        state = torch.empty((n_loc, q), device=dev)
        logp = torch.empty(n_loc, device=dev)
        grad = torch.empty((n_loc, q), device=dev)
        step = torch.full((1,), float(step_size), device=dev)
        total = burn_in + n_mcmc
        acc_prob = torch.zeros(total, device=dev, dtype=torch.float64)
        acc_count = torch.zeros(total, device=dev, dtype=torch.int32)
        n_adapt = int(burn_in * 0.8)
        for it in range(n_adapt):
            eng.hmc_run(x, state, logp, grad, step, it, 1, burn_in, num_leapfrog_steps, seed, init=(it == 0),
                        row_base=lo_r, acc_prob=acc_prob, acc_count=acc_count)
            parallel.all_reduce_sum_(acc_prob[it:it + 1])
            eng.hmc_adapt(step, acc_prob, it, n)
        if burn_in > n_adapt:
            eng.hmc_run(x, state, logp, grad, step, n_adapt, burn_in - n_adapt, burn_in, num_leapfrog_steps, seed,
                        init=(n_adapt == 0), row_base=lo_r, acc_prob=acc_prob, acc_count=acc_count)
        miss_dev = torch.isnan(x)
        k_row_dev = miss_dev.sum(dim=1)
        k_max = k_row_dev.max().reshape(1) if n_loc else torch.zeros(1, dtype=torch.int64, device=dev)
        k_slots = int(parallel.all_reduce_max_(k_max).item())
        slot_dev = (torch.cumsum(miss_dev, dim=1, dtype=torch.int32) - 1).to(torch.int32)
        slot_dev.masked_fill_(~miss_dev, -1)
        per_row = 4 * n_mcmc * (2 * q + max(k_slots, 1) + (p if return_samples else 0))
        rows_chunk = max(bs, int(max_draw_bytes // max(1, per_row)) // bs * bs)
        quantum = 16 * 8 * torch.cuda.get_device_properties(dev).multi_processor_count      # one round of the sampler: 8 waves per CU x 16 chains
        if rows_chunk > quantum:           # whole rounds per launch (a chunk of 1.4 rounds costs nearly 2), kept a multiple of bs
            rows_chunk = max(bs, (rows_chunk // quantum * quantum) // bs * bs)
        means = torch.zeros((n_loc, max(k_slots, 1)), device=dev)
        los = torch.zeros_like(means)
        his = torch.zeros_like(means)
        samples = []
        for s, e, calls in block_plan(lo_r, n_loc, bs, rows_chunk):
            draws = torch.empty((n_mcmc, e - s, q), device=dev)
            eng.hmc_run(x[s:e], state[s:e], logp[s:e], grad[s:e], step, burn_in, n_mcmc, burn_in, num_leapfrog_steps,
                        seed, init=(burn_in == 0), row_base=lo_r + s, acc_count=acc_count, draws=draws)
            for b, be, blk, off in calls:
                if k_slots > 0 or return_samples:
                    cells, full = eng.decode(draws[:, b - s:be - s].contiguous(), seed, STREAM_PREDICT + blk, burn_in=burn_in,
                                             row_base=lo_r + b, slot=slot_dev[b:be].contiguous() if k_slots > 0 else None, k_slots=k_slots,
                                             want_full=return_samples, sign_stride=bs, sign_off=off)
                    if k_slots > 0:
                        mean, lo, hi = eng.row_mean_quantiles(cells.reshape((be - b) * k_slots, n_mcmc), alpha / 2.0, 1.0 - alpha / 2.0)
                        means[b:be] = mean.reshape(be - b, k_slots)
                        los[b:be] = lo.reshape(be - b, k_slots)
                        his[b:be] = hi.reshape(be - b, k_slots)
                    if return_samples:
                        samples.append(full.cpu().numpy())
            del draws
        acc = acc_count[burn_in:].sum().double().reshape(1)
        parallel.all_reduce_sum_(acc)
        self.last_acceptance_rate = float(acc.item()) / max(1, n_mcmc * n)
        print(f"TFP MCMC Acceptance Rate: {self.last_acceptance_rate:.4f}")
        imputed_dev = torch.where(miss_dev, means.gather(1, slot_dev.clamp(min=0).long()), torch.nan_to_num(x, nan=0.0)) \
            if k_slots > 0 else x.clone()
        if parallel.is_dist():
            means, los, his = (parallel.all_gather_rows(a_, n) for a_ in (means, los, his))
            imputed_dev = parallel.all_gather_rows(imputed_dev, n)
        los, his = los.cpu().numpy(), his.cpu().numpy()
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
            flat = np.stack([los[used], his[used]], axis=-1).astype(np.float32)
            pred_interval = np.split(flat, np.cumsum(k_row)[:-1]) if n else []
        if return_samples:
            full = np.concatenate(samples, axis=1)
            if parallel.is_dist():
                full = parallel.all_gather_rows(torch.from_numpy(np.ascontiguousarray(full.transpose(1, 0, 2))).to(dev),
                                                n).cpu().numpy().transpose(1, 0, 2)
            return full, pred_interval
        return imputed_dev.cpu().numpy(), pred_interval
