"""CausalBGM with Bayesian networks (``params['use_bnn'] = True``, the default of the reference's YAML configs and CLI).

Mirrors /root/reference/src/bayesgm/models/causalbgm/base.py with its ``use_bnn`` branches
(:64-72 network construction, :171-173/:205-207/:234-236 KL terms) and networks/bnn.py:4-38.  Same public methods as
the deterministic class; every network call is a Flipout call with fresh noise on the statistics of the batch it is
given, so -- exactly as in the reference -- `predict` depends on ``bs`` (the rows of one block share statistics and
weight perturbations) and `evaluate` treats the panel it is given as one batch.

``params['bnn_norm']`` (build option, default "fixed"): "fixed" = the input BatchNormalization runs in inference mode on its
never-updated moving averages (mean 0 / variance 1), the reading under which the build reproduces the reference's published
training log, acceptance rate and ADRF error (DESIGN_HISTORY.md section 2b); "batch" = it uses the statistics of the batch at hand,
which is what Keras 2.10's documented training-mode propagation implies for the code as written -- a constant counterfactual
treatment column is then normalised away and the ADRF is flat (DESIGN_HISTORY.md section 7).

Stated differences (DESIGN.md "Bayesian nets"): the noise streams are the build's counter-based ones (oracle/bnn.py);
minibatches hold up to 4096 rows per rank (16 and 32 run on the row-tile chains, other sizes on the one-workgroup-per-net kernels; the session is sized for 256 and re-opened for more before its first step, or by params['max_batch']); under torch.distributed every rank normalises with the statistics of ITS rows.
"""
import datetime
import os

import numpy as np
import torch

from .. import diagnostics, host_rng, parallel
from ..bnn_engine import BnnEngine, NETS, flatten_bnn
from ..datasets import Gaussian_sampler
from ..utils import save_data
from .causalbgm import CausalBGM, _DEFAULTS, _glorot, _disc_norm, BNN_NORM_DEFAULT


def _init_bnn(rs, dims):
    """tfp.layers.DenseFlipout defaults (default_mean_field_normal_fn): loc ~ N(0, 0.1^2), untransformed scale
    ~ N(-3, 0.1^2), bias ~ N(0, 0.1^2); BatchNormalization gamma = 1, beta = 0."""
    layers = []
    for i in range(len(dims) - 1):
        layers.append(((0.1 * rs.standard_normal((dims[i], dims[i + 1]))).astype(np.float32),
                       (-3.0 + 0.1 * rs.standard_normal((dims[i], dims[i + 1]))).astype(np.float32),
                       (0.1 * rs.standard_normal(dims[i + 1])).astype(np.float32)))
    return {"gamma": np.ones(dims[0], np.float32), "beta": np.zeros(dims[0], np.float32), "layers": layers}


class CausalBGMBayes(CausalBGM):
    def __init__(self, params, timestamp=None, random_seed=None, device=None):
        self.params = params
        self.timestamp = timestamp
        p = dict(_DEFAULTS)
        p.update(params)
        self._p = p
        for k in ("sigma_v", "sigma_x", "sigma_y"):          # fixed likelihood standard deviations (base.py:161,195,224): > 0
            if k in params and not float(params[k]) > 0.0:
                raise ValueError("params['%s'] must be positive" % k)
        self._bnn_norm = p.get("bnn_norm", BNN_NORM_DEFAULT)
        if "bnn_norm" not in p and self._bnn_norm != "batch":      # a default that is not the reference as written: said once per process
            diagnostics.notice_once("bnn_norm", "params['bnn_norm'] not given: the Bayesian nets' input BatchNormalization runs in inference mode "
                                    "('fixed': the mode that reproduces the published tutorial trace); 'batch' = batch statistics, as "
                                    "networks/bnn.py:25-27 reads (DESIGN_HISTORY.md section 2b)")
        if self._bnn_norm not in ("batch", "fixed"):
            raise ValueError("params['bnn_norm'] must be 'batch' or 'fixed'")
        if self._bnn_norm == "batch":
            import warnings
            warnings.warn("bayesgm_amd: params['bnn_norm'] = 'batch' (input BatchNormalization on batch statistics): a counterfactual "
                          "treatment column that is constant over the batch is normalised away, so ADRF / ITE estimates do not depend on "
                          "the treatment value, and the reference's published results are not reproduced (DESIGN_HISTORY.md sections 2b, 7).",
                          stacklevel=3)
        random_seed = parallel.shared_seed(random_seed)   # None stays None in a single process; one seed for all ranks otherwise
        self._rs = np.random.RandomState(random_seed) if random_seed is not None else np.random.RandomState()
        if random_seed is not None:
            np.random.seed(random_seed)
        self._seed_counter = 0
        self._base_seed = int(random_seed) if random_seed is not None else int(np.random.randint(0, 2 ** 31 - 1))
        self._stream = 0                       # noise call counter of the minibatch steps
        z = list(p["z_dims"])
        q = sum(z)
        self.nets = {
            "g": _init_bnn(self._rs, [q] + list(p["g_units"]) + [p["v_dim"] + 1]),
            "e": _init_bnn(self._rs, [p["v_dim"]] + list(p["e_units"]) + [q]),
            "f": _init_bnn(self._rs, [z[0] + z[1] + 1] + list(p["f_units"]) + [2]),
            "h": _init_bnn(self._rs, [z[0] + z[2]] + list(p["h_units"]) + [2]),
        }
        self.z_sampler = Gaussian_sampler(mean=np.zeros(q), sd=1.0)
        if device is None:
            device = int(os.environ.get("BGM_DEVICE", os.environ.get("LOCAL_RANK", 0)))
        self.engine = BnnEngine(p["v_dim"], z, binary_treatment=p["binary_treatment"], g_units=p["g_units"], e_units=p["e_units"],
                                f_units=p["f_units"], h_units=p["h_units"], kl_weight=p["kl_weight"], max_batch=int(params.get("max_batch", 256)),
                                norm_mode={"batch": 0, "fixed": 1}[self._bnn_norm], device=device,
                                sigma_v=params.get("sigma_v"), sigma_x=params.get("sigma_x"), sigma_y=params.get("sigma_y"))
        self.engine.set_disc_norm(_disc_norm(p))
        self.engine.begin(self.nets)
        # params['mh_precision'] (build option): arithmetic of the posterior-sampling kernels of predict / get_log_posterior /
        # metropolis_hastings_sampler.  'fp32' (default) | 'f16x3' (split precision on the fp16 matrix pipe: csrc/bnx_kernels.h, within the
        # fp32 kernels' own distance of float64, ~1.9x faster); fit, egm_init and evaluate stay fp32.  There is no bf16 form of this family.
        if p.get("mh_precision", "fp32") not in ("fp32", "f16x3"):
            raise ValueError("params['mh_precision'] must be 'fp32' or 'f16x3' for use_bnn=True models")
        if p.get("mh_precision", "fp32") != "fp32":
            self.engine.set_precision(p["mh_precision"])
        if self.timestamp is None:
            self.timestamp = datetime.datetime.now().astimezone().strftime('%Y%m%d_%H%M%S')
        self.checkpoint_path = "{}/checkpoints/{}/{}".format(params['output_dir'], params['dataset'], self.timestamp)
        if p['save_model'] and not os.path.exists(self.checkpoint_path):
            os.makedirs(self.checkpoint_path, exist_ok=True)
        self.save_dir = "{}/results/{}/{}".format(params['output_dir'], params['dataset'], self.timestamp)
        if p['save_res'] and not os.path.exists(self.save_dir):
            os.makedirs(self.save_dir, exist_ok=True)
        self.data_z = None
        self.last_acceptance_rate = None
        self._restore_latest()

    # ------------------------------------------------------------------ plumbing
    def _noise_seed(self, per_rank=False):
        """Noise key of the network calls.  The EGM warm start and evaluate are replicated / reduced over ranks and use one
        key; the data-parallel minibatch steps of fit draw different perturbations on every rank (per_rank)."""
        return (self._base_seed * 2654435761 + (97 * parallel.rank() if per_rank else 0)) & 0x7FFFFFFFFFFFFFFF

    def _streams(self, n):
        s = self._stream
        self._stream = (self._stream + n) & 0x3FFFFFFF
        return s

    def _pull_weights(self, which=None):
        self.nets = self.engine.split(self.engine.read(0))

    def _push_weights(self, which=None):
        self.engine.write(np.concatenate([flatten_bnn(self.nets[k]) for k in NETS]))

    def set_weights(self, **nets):
        """Install network parameters ({"gamma", "beta", "layers": [(loc, rho, bias), ...]} per net)."""
        for k, v in nets.items():
            self.nets[k] = {"gamma": np.asarray(v["gamma"], np.float32), "beta": np.asarray(v["beta"], np.float32),
                            "layers": [tuple(np.asarray(a, np.float32) for a in L) for L in v["layers"]]}
        self._push_weights()

    def save_checkpoint(self, epoch):
        """ckpt_manager.save(epoch): parameters and the Adam slots of the session (g / f / h_optimizer), the noise-stream counter,
        the latent table when written from inside `fit`; at most 5 kept (base.py:112-128, 527-529)."""
        flat = dict(theta=self.engine.read(0), opt_m=self.engine.read(2), opt_v=self.engine.read(3),
                    stream=np.array([self._stream], np.int64))
        if self.data_z is not None:
            flat["data_z"] = self.data_z.cpu().numpy()
        path = self.ckpt_manager.save("ckpt-%s.npz" % epoch, flat)
        print('Saving checkpoint for epoch {} at {}'.format(epoch, path))
        return path

    def load_checkpoint(self, path):
        d = np.load(path)
        self.engine.write(d["theta"])
        if "opt_m" in d.files:
            self.engine.write(d["opt_m"], 2)
            self.engine.write(d["opt_v"], 3)
            self._stream = int(d["stream"][0])
        self._pull_weights()

    # ------------------------------------------------------------------ EGM
    def egm_init(self, data, egm_n_iter=30000, batch_size=32, egm_batches_per_eval=500, verbose=1):
        """EGM warm start (base.py:380-431) with Bayesian g, e, f, h: csrc/bnn_egm_kernels.h, one launch per step."""
        data_x, data_y, data_v = data
        n = len(data_x)
        eng = self.engine
        dev = eng.device
        p_ = self._p
        # Under torch.distributed the warm start is data parallel (north_star's collective, base.py:305-377): every rank holds ITS rows
        # of the panel, runs each step on its batch_size // world share of the global minibatch, and the discriminator / generator
        # gradients are all-reduced (RCCL) before the Adam step, so all ranks hold identical networks throughout.
        world, rank = parallel.world_size(), parallel.rank()
        lo_r, hi_r = parallel.shard_range(n)
        n_loc, b_loc = hi_r - lo_r, batch_size // world
        if world > 1 and b_loc < 2:
            raise ValueError("egm_init under torch.distributed: batch_size // world_size must be at least 2")
        # tests: ONE process steps on the minibatches a `_egm_emulate_world`-rank run forms (the ranks' shares side by side)
        emu = int(getattr(self, "_egm_emulate_world", 0)) if world == 1 else 0
        if emu > 1:
            b_loc = (batch_size // emu) * emu
        eng.ensure_max_batch(b_loc)          # (base.py:380: any batch_size; the session is sized here, before its first step)
        xd, yd, vd = self._dev(data_x[lo_r:hi_r]).reshape(-1), self._dev(data_y[lo_r:hi_r]).reshape(-1), self._dev(data_v[lo_r:hi_r])
        q = sum(p_["z_dims"])
        dims = [q] + list(p_["dz_units"]) + [1]
        dz = {"W": [_glorot(self._rs, dims[i], dims[i + 1]) for i in range(len(dims) - 1)],       # networks/base.py:338-363
              "b": [np.zeros(dims[i + 1], np.float32) for i in range(len(dims) - 1)],
              "gamma": [np.ones(dims[i + 1], np.float32) for i in range(len(dims) - 2)],
              "beta": [np.zeros(dims[i + 1], np.float32) for i in range(len(dims) - 2)]}
        eng.egm_begin(dz, b_loc, p_["lr"], p_["use_z_rec"])
        out_d = torch.zeros(2, device=dev)
        out_g = torch.zeros(6, device=dev)
        if world > 1:
            eng.egm_set_share(rank * b_loc)
            n_gen, n_dz = eng.egm_sizes()
            buf_g, buf_d = torch.empty(n_gen, device=dev), torch.empty(n_dz, device=dev)
        egm_log = []
        if verbose:
            print('EGM Initialization Starts ...')
        g_d_freq = int(p_['g_d_freq'])
        steps = g_d_freq + 1
        seed = self._noise_seed()
        try:
            # Blocks of iterations up to and including the next evaluation point.  The host random numbers of block k + 1 are drawn
            # (reference order) on a worker thread, on a PRIVATE copy of the generator state, while the GPU runs block k and its
            # evaluation pass; np.random's global state moves only here, at hand-over, to where the sequential loop would have left
            # it (host_rng.EgmDrawPipeline: a block is redrawn if anything consumed np.random in between).
            blocks, bi = [], 0
            while bi <= egm_n_iter:
                stop = min(egm_n_iter, (bi // egm_batches_per_eval + 1) * egm_batches_per_eval if bi % egm_batches_per_eval else bi)
                blocks.append((bi, stop))
                bi = stop + 1

            pipe = host_rng.EgmDrawPipeline(n, batch_size, q, g_d_freq)
            pipe.request(blocks[0][1] - blocks[0][0] + 1)
            for kb, (batch_iter, stop) in enumerate(blocks):
                n_it = stop - batch_iter + 1
                idx_h, z_h, eps3 = pipe.take(blocks[kb + 1][1] - blocks[kb + 1][0] + 1 if kb + 1 < len(blocks) else 0)
                eps_h = eps3[:, :, 0]
                if world > 1:
                    idx_h, z_h = host_rng.egm_rank_share(idx_h, z_h, n, n_loc, b_loc, rank)
                elif emu > 1:
                    parts = []
                    for r in range(emu):
                        lo_e, hi_e = parallel.shard_range(n, r, emu)
                        parts.append(host_rng.egm_rank_share(idx_h, z_h, n, hi_e - lo_e, b_loc // emu, r)[0] + lo_e)
                    idx_h, z_h = np.ascontiguousarray(np.concatenate(parts, axis=2)), np.ascontiguousarray(z_h[:, :, :b_loc])
                idx_d, z_d = torch.from_numpy(idx_h).to(dev), torch.from_numpy(z_h).to(dev)
                for i in range(n_it if world == 1 else 0):
                    for j in range(g_d_freq):
                        eng.egm_disc_step(z_d[i, j], idx_d[i, j], vd, eps_h[i, j], seed, self._streams(1), out=out_d)
                    eng.egm_gen_step(z_d[i, g_d_freq], idx_d[i, g_d_freq], vd, xd, yd, seed, self._streams(9), out=out_g)
                for i in range(n_it if world > 1 else 0):       # same (seed, stream) on every rank: one weight perturbation per global step
                    for j in range(g_d_freq):
                        eng.egm_disc_step(z_d[i, j], idx_d[i, j], vd, eps_h[i, j], seed, self._streams(1), apply=False, out=out_d)
                        eng.egm_grad(1, 1.0 / world, buf_d)
                        parallel.all_reduce_sum_(buf_d)                      # dz gradient of the global minibatch
                        eng.egm_apply(1, buf_d)
                    eng.egm_gen_step(z_d[i, g_d_freq], idx_d[i, g_d_freq], vd, xd, yd, seed, self._streams(9), apply=False, out=out_g)
                    eng.egm_grad(0, 1.0 / world, buf_g)
                    parallel.all_reduce_sum_(buf_g)                          # fused g | e | f | h gradient (data terms + KL)
                    eng.egm_apply(0, buf_g)
                batch_iter = stop
                if batch_iter % egm_batches_per_eval == 0:
                    if world > 1:                                            # the log line shows means over the global minibatch
                        parallel.all_reduce_sum_(out_g); parallel.all_reduce_sum_(out_d)
                        out_g /= world; out_d /= world
                    lg, ld = out_g.cpu().numpy(), out_d.cpu().numpy()
                    egm_log.append((batch_iter, float(lg[2])))          # l2_loss_z of this log line (diagnostics.py)
                    if verbose:
                        print('EGM Initialization Iter [%d] : e_loss_adv [%.4f], l2_loss_v [%.4f], l2_loss_z [%.4f], '
                              'l2_loss_x [%.4f], l2_loss_y [%.4f], g_e_loss [%.4f], dz_loss [%.4f], d_loss [%.4f]'
                              % (batch_iter, lg[0], lg[1], lg[2], lg[3], lg[4], lg[5], ld[0], ld[1]))
                    causal_pre, mse_x, mse_y, mse_v = self.evaluate(data=data)
                    if self._p['save_res'] and parallel.rank() == 0:
                        save_data('{}/causal_pre_egm_init_iter-{}.txt'.format(self.save_dir, batch_iter), causal_pre)
        finally:
            if 'pipe' in locals():
                pipe.close()
            eng.egm_end()
            self._pull_weights()
        # (the thresholds describe the full-length warm start: a short one is merely unconverged and gets no diagnosis)
        self._egm_late_l2z = diagnostics.late_l2_loss_z([a for a, _ in egm_log], [b for _, b in egm_log], egm_n_iter) if (egm_log and egm_n_iter >= diagnostics.MIN_EGM_ITER) else None
        self._second_optimum_warned = False
        if verbose:
            print('EGM Initialization Ends.')

    # ------------------------------------------------------------------ fit
    def fit(self, data, epochs=100, epochs_per_eval=5, batch_size=32, startoff=0, use_egm_init=True,
            egm_n_iter=30000, egm_batches_per_eval=500, save_format='txt', verbose=1, z_adam=None, host_loop=False, dp_comm=None):
        """Iterative theta / Z updates (base.py:434-532) with the KL terms of the Bayesian nets.  ``batch_size`` is the
        GLOBAL minibatch (any size as in base.py:434; up to 4096 rows per rank, 16 / 32 on the row-tile chains, other sizes on the
        one-workgroup-per-net kernels); under torch.distributed rows are sharded, the g | h | f gradients all-reduced.
        ``host_loop=False`` (single process): one library call per epoch (bgm_bnn_fit_epoch), the latent phase of a minibatch beside
        the chains of the next on a second stream; ``True``: the per-minibatch calls from Python (same results)."""
        if z_adam is None:
            z_adam = "replay"
            diagnostics.notice_once("z_adam", "fit(z_adam=...) not given: the latent Adam runs in its replayed form ('replay': equal to Keras' "
                                    "dense-decay sweep up to fp32 rounding); 'dense' executes the sweep as the reference does (DESIGN_HISTORY.md section 4d)")
        self.engine.ensure_max_batch(max(2, batch_size // parallel.world_size()))
        if use_egm_init:
            self.egm_init(data, egm_n_iter=egm_n_iter, batch_size=batch_size,
                          egm_batches_per_eval=egm_batches_per_eval, verbose=verbose)
        data_x, data_y, data_v = data
        n_total = len(data_x)
        if self._p['save_res'] and parallel.rank() == 0:
            with open('{}/params.txt'.format(self.save_dir), 'w') as f_params:
                f_params.write(str(self.params))
        eng = self.engine
        q = eng.q
        lo_r, hi_r = parallel.shard_range(n_total)
        n_loc = hi_r - lo_r
        world = parallel.world_size()
        dev = eng.device
        x = self._dev(data_x[lo_r:hi_r]).reshape(-1)
        y = self._dev(data_y[lo_r:hi_r]).reshape(-1)
        v = self._dev(data_v[lo_r:hi_r])
        seed = self._noise_seed(per_rank=True)
        if use_egm_init:
            if verbose:
                print('Initialize latent variables Z with e(V)...')
            self.data_z, _, _ = eng.evaluate(None, None, v, None, seed=seed, stream_id=self._streams(1), want_sums=False,
                                             want_effects=False)                        # base.py:479: one call of e on the panel
        else:
            if verbose:
                print('Random initialization of latent variables Z...')
            self.data_z = self._dev(np.random.normal(0, 1, size=(n_total, q)).astype('float32')[lo_r:hi_r])   # base.py:482
        zm = torch.zeros_like(self.data_z)
        zv = torch.zeros_like(self.data_z)
        b_loc = max(2, batch_size // world)
        # the same number of steps and batch sizes on every rank (one all-reduce per step): see CausalBGM.fit
        n_use = n_total // world if world > 1 else n_loc
        grad = torch.empty(eng.n_params, device=dev, dtype=torch.float32) if world > 1 else None
        out_t = torch.zeros(8, device=dev)
        out_z = torch.zeros(4, device=dev)
        if z_adam not in ("dense", "lazy", "replay"):
            raise ValueError("z_adam must be 'replay', 'dense' or 'lazy'")
        lazy = {"dense": 0, "lazy": 1, "replay": 2}[z_adam]      # see CausalBGM.fit
        replay = (lazy == 2)
        lr_z = self._p['lr_z']
        best_loss = np.inf
        if dp_comm is None and world > 1 and not host_loop:      # see CausalBGM.fit
            dp_comm = parallel.fit_comm(dev)
        self.last_fit_path = ("library_epoch_dp (RCCL all-reduce inside bgm_bnn_fit_epoch_dp, %d ranks)" % dp_comm.world if dp_comm is not None
                              else "library_epoch" if (world == 1 and not host_loop) else "host_loop")
        if verbose:
            print('Iterative Updating Starts ...')
        try:
            for epoch in range(epochs + 1):
                sample_idx = torch.from_numpy(np.random.choice(n_loc, n_loc, replace=False).astype(np.int32)).to(dev)
                in_library = (world == 1 and not host_loop) or dp_comm is not None      # the minibatch loop inside the library
                if dp_comm is not None:                          # (bgm_bnn_fit_epoch_dp: one ncclAllReduce per step, enqueued from C++)
                    done = eng.fit_epoch_dp(dp_comm, x, y, v, self.data_z, zm, zv, sample_idx[:n_use], b_loc, self._p['lr_theta'], lr_z, lazy,
                                            seed, self._stream, out_t, out_z)
                    self._streams(3 * done)
                elif in_library:                                 # (bgm_bnn_fit_epoch)
                    done = eng.fit_epoch(x, y, v, self.data_z, zm, zv, sample_idx[:n_use], b_loc, self._p['lr_theta'], lr_z, lazy, seed,
                                         self._stream, out_t, out_z)
                    self._streams(3 * done)
                for i in (() if in_library else range(0, n_use, b_loc)):
                    idx = sample_idx[i:min(i + b_loc, n_use)]
                    if idx.numel() < 2:
                        continue                      # batch statistics need two rows (the same decision on every rank)
                    bg = int(idx.numel()) * world
                    s0 = self._streams(3)
                    if replay:
                        eng.z_sync(self.data_z, zm, zv, idx, lr_z)
                    if world > 1:
                        eng.theta_step(self.data_z, idx, x, y, v, self._p['lr_theta'], seed, s0, apply=False, batch_global=bg, out=out_t)
                        eng.grad_exchange(grad, False)
                        parallel.all_reduce_sum_(grad)
                        eng.grad_exchange(grad, True)
                        eng.theta_apply(self._p['lr_theta'])
                    else:
                        eng.theta_step(self.data_z, idx, x, y, v, self._p['lr_theta'], seed, s0, apply=True, out=out_t)
                    eng.z_step(x, y, v, self.data_z, zm, zv, idx, self._p['lr_z'], seed, s0 + 1, lazy=lazy, batch_global=bg, out=out_z)
                if verbose:
                    lt, lz = out_t.cpu().numpy(), out_z.cpu().numpy()
                    print('Epoch [%d/%d]: loss_px_z [%.4f], loss_mse_x [%.4f], loss_py_z [%.4f], loss_mse_y [%.4f], loss_pv_z [%.4f], '
                          'loss_mse_v [%.4f], loss_postrior_z [%.4f]' % (epoch, epochs, lt[2], lt[3], lt[4], lt[5], lt[0], lt[1], lz[0]))
                if replay and (epoch % epochs_per_eval == 0 or epoch == epochs):
                    eng.z_sync(self.data_z, zm, zv, None, lr_z)                      # flush: evaluate / checkpoints read the whole table
                if epoch % epochs_per_eval == 0:
                    causal_pre, mse_x, mse_y, mse_v = self._evaluate_dev(x, y, v, self.data_z, n_total, lo_r)
                    if self.params.get('second_optimum_check', True):
                        self._second_optimum_warned = diagnostics.warn_if_second_optimum(getattr(self, '_egm_late_l2z', None), float(mse_v),
                                                                                         getattr(self, '_second_optimum_warned', False),
                                                                                         v_var=self._panel_v_var(v))
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
        finally:
            if replay:          # an interrupted fit must not leave rows of data_z with unreplayed steps
                try:
                    eng.z_sync(self.data_z, zm, zv, None, lr_z)
                except Exception:
                    pass
        self._pull_weights()

    # ------------------------------------------------------------------ evaluate
    def _evaluate_dev(self, x, y, v, z, n_total, lo_r, nb_intervals=200, x_full=None):
        eng = self.engine
        seed = self._noise_seed()
        if self._p['binary_treatment']:
            z, sums, ite = eng.evaluate(x, y, v, z, seed=seed, stream_id=self._streams(3))
            parallel.all_reduce_sum_(sums)
            ite = parallel.all_gather_rows(ite.reshape(-1, 1), n_total)
            s = sums.cpu().numpy()
            return (ite.cpu().numpy(), np.float32(s[1] / n_total), np.float32(s[2] / n_total),
                    np.float32(s[0] / (n_total * eng.v_dim)))
        xs_all = parallel.all_gather_rows(x.reshape(-1, 1), n_total).cpu().numpy() if x_full is None else x_full
        x_min = self._percentile_nearest(xs_all, 5.0)
        x_max = self._percentile_nearest(xs_all, 95.0)
        x_values = np.linspace(x_min, x_max, nb_intervals).astype(np.float32)
        z, sums, dose = eng.evaluate(x, y, v, z, x_values=x_values, seed=seed, stream_id=self._streams(1 + nb_intervals))
        parallel.all_reduce_sum_(sums)
        parallel.all_reduce_sum_(dose)
        s = sums.cpu().numpy()
        return ((dose / n_total).float().cpu().numpy(), np.float32(s[1] / n_total), np.float32(s[2] / n_total),
                np.float32(s[0] / (n_total * eng.v_dim)))

    def evaluate(self, data, data_z=None, nb_intervals=200):
        """(causal_pre, mse_x, mse_y, mse_v) (base.py:534-570); data_z=None -> one call of e_net on the panel."""
        data_x, data_y, data_v = data
        n_total = len(data_x)
        lo_r, hi_r = parallel.shard_range(n_total)
        x = self._dev(data_x[lo_r:hi_r]).reshape(-1)
        y = self._dev(data_y[lo_r:hi_r]).reshape(-1)
        v = self._dev(data_v[lo_r:hi_r])
        z = None
        if data_z is not None:
            z = self._dev(data_z[lo_r:hi_r]) if len(data_z) == n_total else self._dev(data_z)
        return self._evaluate_dev(x, y, v, z, n_total, lo_r, nb_intervals)

    # ------------------------------------------------------------------ predict
    def _run_chains(self, x, y, v, bs, burn_in, n_keep, q_sd, seed, row_base, block0, adaptive, initial_q_sd=1.0, target=0.25,
                    tol=0.05, adj_int=50, window=100, block_row0=0, block_rows_global=None, **outs):
        """All blocks of the given rows in lock step.  Every block is its own sampler run of the reference
        (metropolis_hastings_sampler is called once per block, base.py:640-645), so an adaptive proposal scale is kept PER
        BLOCK: at counter = 50, 100, ... < burn_in the acceptance rate of the block's last `window` iterations decides
        q_sd *= 0.9 / 1.1 (base.py:873-892).  Returns (final states, accepted proposals of the last <= 100 iterations, that
        window's length)."""
        eng = self.engine
        dev = eng.device
        n = x.shape[0]
        n_blocks = (n + bs - 1) // bs
        rows_b = np.minimum(bs, n - bs * np.arange(n_blocks)).astype(np.float64)
        # ``block_rows_global`` (with ``block_row0``): the rows given are this rank's share of ONE block of that many rows -- the
        # acceptance counts that steer the block's proposal scale and the final report are then summed over the ranks
        shared = block_rows_global is not None
        if shared:
            rows_b = np.array([float(block_rows_global)])
        state = torch.empty((n, eng.q), device=dev, dtype=torch.float32)
        total = burn_in + n_keep
        tail = min(100, total)
        sd = torch.full((n_blocks,), float(initial_q_sd if adaptive else q_sd), device=dev, dtype=torch.float32)
        bounds = {total, total - tail}
        if adaptive:
            bounds |= {c + 1 for c in range(adj_int, burn_in, adj_int)}     # adjust after iteration counter = 50, 100, ...
        hist = np.zeros((0, n_blocks))                                       # accepted proposals per iteration and block
        it, acc_tail = 0, 0
        for b in sorted(b for b in bounds if 0 < b <= total):
            seg = b - it
            acc = torch.zeros((seg, n_blocks), device=dev, dtype=torch.int32)
            eng.mh_run(x, y, v, state, bs, it, seg, burn_in, 1.0, seed, init=(it == 0), row_base=row_base, block0=block0,
                       n_keep=n_keep, q_sd_blocks=sd, acc_blocks=acc, block_row0=block_row0, **outs)
            if shared:
                parallel.all_reduce_sum_(acc)
            a = acc.cpu().numpy().astype(np.float64)
            if it >= total - tail:
                acc_tail += int(a.sum())
            if adaptive and b <= burn_in and (b - 1) % adj_int == 0:
                hist = np.concatenate([hist, a])[-window:]
                rate = hist.sum(axis=0) / (len(hist) * rows_b)
                f = np.where(rate < target - tol, 0.9, np.where(rate > target + tol, 1.1, 1.0))
                sd = sd * torch.from_numpy(f.astype(np.float32)).to(dev)
            elif adaptive:
                hist = np.concatenate([hist, a])[-window:]
            it = b
        self.last_q_sd = sd.cpu().numpy()
        return state, acc_tail, tail

    def predict(self, data, alpha=0.01, n_mcmc=3000, burn_in=5000, x_values=None, q_sd=1.0, sample_y=True,
                bs=10000, verbose=1):
        """Causal effects with posterior intervals (base.py:573-668).  With Bayesian nets the rows of one block of ``bs``
        rows share their input statistics and weight perturbations, as in the reference; all blocks advance together."""
        assert 0 < alpha < 1, "The significance level 'alpha' must be greater than 0 and less than 1."
        parallel.check_n_mcmc(n_mcmc)
        binary = bool(self._p['binary_treatment'])
        if not binary and x_values is None:
            raise ValueError("For continuous treatment, 'x_values' must not be None. Provide a list or a single treatment value.")
        if x_values is not None:
            x_values = np.array([x_values], dtype=float) if np.isscalar(x_values) else np.array(x_values, dtype=float)
        data_x, data_y, data_v = data
        n_test = len(data_x)
        bs = max(2, int(bs))
        # shard by whole blocks so that a block's statistics never span ranks
        n_blocks = (n_test + bs - 1) // bs
        b_lo, b_hi = parallel.shard_range(n_blocks)
        lo_r, hi_r = min(n_test, b_lo * bs), min(n_test, b_hi * bs)
        n_loc = hi_r - lo_r
        eng = self.engine
        dev = eng.device
        adaptive = (q_sd is None) or (q_sd <= 0)
        seed = self._next_seed()
        if verbose:
            print('MCMC Latent Variable Sampling ...')
        x = self._dev(data_x[lo_r:hi_r]).reshape(-1)
        y = self._dev(data_y[lo_r:hi_r]).reshape(-1)
        v = self._dev(data_v[lo_r:hi_r])
        acc_tail, tail = 0, min(100, burn_in + n_mcmc)        # acceptance window of the final report (base.py:899-901)
        if binary:
            max_rows = max(bs, int((32 << 30) // (4 * max(1, n_mcmc))) // bs * bs)      # whole blocks, draw matrix <= ~32 GiB
            means, los, his = [], [], []
            for s in range(0, n_loc, max_rows):
                e = min(s + max_rows, n_loc)
                ite = torch.empty((e - s, n_mcmc), device=dev, dtype=torch.float32)
                _, a_, tail = self._run_chains(x[s:e], y[s:e], v[s:e], bs, burn_in, n_mcmc, q_sd, seed, lo_r + s, b_lo + s // bs,
                                               adaptive, effect=2, sample_y=sample_y, ite=ite)
                acc_tail += a_
                mean, lo, hi = eng.row_mean_quantiles(ite, alpha / 2, 1 - alpha / 2)
                means.append(mean); los.append(lo); his.append(hi)
            cat = lambda ts: torch.cat(ts) if ts else torch.empty(0, device=dev)
            mean = parallel.all_gather_rows_var(cat(means).reshape(-1, 1)).reshape(-1)
            lo = parallel.all_gather_rows_var(cat(los).reshape(-1, 1)).reshape(-1)
            hi = parallel.all_gather_rows_var(cat(his).reshape(-1, 1)).reshape(-1)
            self._report_acceptance(float(acc_tail), tail, n_test, verbose)
            return mean.cpu().numpy(), torch.stack([lo, hi], dim=1).cpu().numpy()
        xv = self._dev(x_values.astype(np.float32))
        sums = torch.zeros((len(x_values), n_mcmc), device=dev, dtype=torch.float64)
        if n_loc > 0:
            _, acc_tail, tail = self._run_chains(x, y, v, bs, burn_in, n_mcmc, q_sd, seed, lo_r, b_lo, adaptive, effect=1,
                                                 sample_y=sample_y, x_values=xv, adrf_sum=sums)
        parallel.all_reduce_sum_(sums)                        # adrf_draw_sums (base.py:660)
        causal_effects = (sums / float(n_test)).float().contiguous()
        adrf, lo, hi = eng.row_mean_quantiles(causal_effects, alpha / 2, 1 - alpha / 2)
        self._report_acceptance(float(acc_tail), tail, n_test, verbose)
        return adrf.cpu().numpy(), torch.stack([lo, hi], dim=1).cpu().numpy()

    def metropolis_hastings_sampler(self, data, initial_q_sd=1.0, q_sd=None, burn_in=5000, n_keep=3000,
                                    target_acceptance_rate=0.25, tolerance=0.05, adjustment_interval=50,
                                    adaptive_sd=None, window_size=100):
        """Posterior samples of Z, shape (n_keep, n, q) (base.py:820-904): the rows given are ONE block."""
        data_x, data_y, data_v = data
        if adaptive_sd is None:
            adaptive_sd = (q_sd is None or q_sd <= 0)
        n = len(data_x)
        draws = torch.empty((n_keep, n, self.engine.q), device=self.engine.device, dtype=torch.float32)
        _, acc_tail, tail = self._run_chains(self._dev(data_x).reshape(-1), self._dev(data_y).reshape(-1), self._dev(data_v), max(2, n),
                                             burn_in, n_keep, q_sd, self._next_seed(), 0, 0, adaptive_sd, initial_q_sd=initial_q_sd,
                                             target=target_acceptance_rate, tol=tolerance, adj_int=adjustment_interval, window=window_size, draws=draws)
        self.last_acceptance_rate = acc_tail / float(tail * n)
        print(f"Final MCMC Acceptance Rate: {self.last_acceptance_rate:.4f}")
        return draws.cpu().numpy()

    def infer_from_latent_posterior(self, data_posterior_z, x_values=None, sample_y=True, eps=1e-6, seed=None):
        """Causal effects from posterior draws of Z, shape (n_keep, n, q) (base.py:671-763): the n rows are ONE block, as in the
        reference's call; binary -> ITE draws (n_keep, n); continuous -> ADRF draws (len(x_values), n_keep)."""
        if not self._p["binary_treatment"] and x_values is None:
            raise ValueError("For continuous treatment, `x_values` must not be None. Provide a list or numpy array.")
        draws = self._dev(data_posterior_z)
        out = self.engine.effects(draws, max(2, draws.shape[1]), self._next_seed() if seed is None else seed, x_values=x_values,
                                  sample_y=sample_y)
        return out.cpu().numpy()

    def get_log_posterior(self, data_x, data_y, data_v, data_z, eps=1e-6):
        """log p(z | x, y, v) + const, shape (n,) (base.py:765-817): one noisy call of g, h, f on the rows given."""
        n = len(data_x)
        return self.engine.logpost(self._dev(data_x).reshape(-1), self._dev(data_y).reshape(-1), self._dev(data_v),
                                   self._dev(data_z), max(2, n), self._next_seed(), 0).cpu().numpy()
