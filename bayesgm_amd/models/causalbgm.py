"""CausalBGM -- host-side mirror of the reference class (same constructor, method
names, argument meaning, return values, error behaviour and output files), driving the
gfx950 kernels of libbgm_hip.so.

Mirrors /root/reference/src/bayesgm/models/causalbgm/base.py:
    __init__ :56-128   get_config :130   fit :434   evaluate :535   predict :573
    metropolis_hastings_sampler :820   infer_from_latent_posterior :672
This class holds the deterministic networks (``use_bnn=False``); with ``use_bnn=True`` (the
reference's default) the constructor returns the Bayesian-network subclass of causalbgm_bnn.py.
"""
import datetime
import os

import numpy as np
import torch

# Default BatchNormalization reading of the layers the reference calls without `training=` (Discriminator, networks/base.py:378;
# BayesianFullyConnectedNet, networks/bnn.py:26): inference mode.  This is the reading under which the build reproduces the
# training log, acceptance rate and ADRF error the reference published (tests/golden/tutorial_trace.json, DESIGN_HISTORY.md section 2b);
# "batch" (batch statistics, what Keras 2.10's documented training-mode propagation implies for the code as written) stays
# available as params['disc_norm'] / params['bnn_norm'] = "batch".
DISC_NORM_DEFAULT = "fixed"
BNN_NORM_DEFAULT = "fixed"

from .. import _lib, diagnostics, host_rng, parallel
from ..engine import CausalEngine
from ..datasets import Gaussian_sampler
from ..utils import save_data
from ._checkpoint import CheckpointManager

_DEFAULTS = dict(use_bnn=True, g_units=[64] * 5, e_units=[64] * 5, f_units=[64, 32, 8], h_units=[64, 32, 8],
                 dz_units=[64, 32, 8], lr=0.0002, lr_theta=0.0001, lr_z=0.0001, g_d_freq=5, save_model=False,
                 save_res=True, kl_weight=0.0001, use_z_rec=True)


def _glorot(rs, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rs.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32)


def _init_mlp(rs, dims):
    """Keras Dense defaults: glorot-uniform kernel, zero bias (networks/base.py:17-26)."""
    return [(_glorot(rs, dims[i], dims[i + 1]), np.zeros(dims[i + 1], np.float32)) for i in range(len(dims) - 1)]


def _disc_norm(p):
    """params['disc_norm'] (build option): BatchNormalization mode of the EGM discriminators, "batch" | "fixed"
    (include/bgm_hip.h bgm_set_disc_norm, DESIGN_HISTORY.md section 2b)."""
    mode = p.get("disc_norm", DISC_NORM_DEFAULT)
    if mode not in ("batch", "fixed"):
        raise ValueError("params['disc_norm'] must be 'batch' or 'fixed'")
    if "disc_norm" not in p and mode != "batch":      # a default that is not the reference as written: said once per process
        diagnostics.notice_once("disc_norm", "params['disc_norm'] not given: the EGM discriminators normalise in inference mode ('fixed'); "
                                "'batch' runs their BatchNormalization on batch statistics as networks/base.py:364-379 reads (DESIGN_HISTORY.md section 2b)")
    return mode


class CausalBGM(object):
    def __new__(cls, params, *args, **kwargs):
        # params['use_bnn'] (default True, base.py:64): the Bayesian-network model lives in causalbgm_bnn.py
        if cls is CausalBGM and dict(_DEFAULTS, **params)["use_bnn"]:
            from .causalbgm_bnn import CausalBGMBayes
            return super().__new__(CausalBGMBayes)
        return super().__new__(cls)

    def __init__(self, params, timestamp=None, random_seed=None, device=None):
        self.params = params
        self.timestamp = timestamp
        p = dict(_DEFAULTS)
        p.update(params)
        self._p = p
        random_seed = parallel.shared_seed(random_seed)   # None stays None in a single process; one seed for all ranks otherwise
        self._rs = np.random.RandomState(random_seed) if random_seed is not None else np.random.RandomState()
        if random_seed is not None:
            np.random.seed(random_seed)
        self._seed_counter = 0
        self._base_seed = int(random_seed) if random_seed is not None else int(np.random.randint(0, 2 ** 31 - 1))
        z = list(p["z_dims"])
        q = sum(z)
        self.nets = {
            "g": _init_mlp(self._rs, [q] + list(p["g_units"]) + [p["v_dim"] + 1]),
            "e": _init_mlp(self._rs, [p["v_dim"]] + list(p["e_units"]) + [q]),
            "f": _init_mlp(self._rs, [z[0] + z[1] + 1] + list(p["f_units"]) + [2]),
            "h": _init_mlp(self._rs, [z[0] + z[2]] + list(p["h_units"]) + [2]),
        }
        self.z_sampler = Gaussian_sampler(mean=np.zeros(q), sd=1.0)
        if device is None:
            device = int(os.environ.get("BGM_DEVICE", os.environ.get("LOCAL_RANK", 0)))   # BGM_DEVICE: dev aid
        self.engine = CausalEngine(p["v_dim"], z, binary_treatment=p["binary_treatment"], g_units=p["g_units"],
                                   f_units=p["f_units"], h_units=p["h_units"], e_units=p["e_units"],
                                   sigma_v=params.get("sigma_v"), sigma_x=params.get("sigma_x"),
                                   sigma_y=params.get("sigma_y"), device=device)
        self.engine.set_disc_norm(_disc_norm(p))
        # params['mh_precision'] (build option): arithmetic of the posterior-sampling kernels of predict /
        # metropolis_hastings_sampler / get_log_posterior: "fp32" (default, the reference's arithmetic) or "bf16x3" (split
        # precision on the bf16 matrix pipe, DESIGN_HISTORY.md section 4b)
        if p.get("mh_precision", "fp32") not in ("fp32", "bf16x3", "f16x3"):
            raise ValueError("params['mh_precision'] must be 'fp32', 'bf16x3' or 'f16x3'")
        self.engine.set_precision(p.get("mh_precision", "fp32"))
        # params['outcome_cache'] (build option, default True): predict's ADRF sampler evaluates the outcome net only for the chains that
        # moved since the last retained draw (identical results; 'wave' = per 16-chain tile, the round-4 form; False = evaluate f at
        # every retained draw as the reference).  params['event_budget_mb']: bound on the event buffers of one segment (default 8192).
        self.engine.set_outcome_cache(p.get("outcome_cache", True))
        if p.get("event_budget_mb"):
            self.engine.set_event_budget(int(p["event_budget_mb"]) << 20)
        self._push_weights()
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
        self._fit_live = None            # (zm, zv) of the running fit, for checkpoints
        self._restored_opt = None        # optimizer slots of a restored checkpoint, installed by the next fit
        self._restore_latest()

    def _restore_latest(self):
        """tf.train.CheckpointManager(..., max_to_keep=5) + restore of the latest checkpoint at construction (base.py:124-128)."""
        self.ckpt_manager = CheckpointManager(self.checkpoint_path, max_to_keep=5)
        latest = self.ckpt_manager.latest_checkpoint
        if latest:
            self.load_checkpoint(latest)
            print('Latest checkpoint restored!!')

    # ------------------------------------------------------------------ plumbing
    def get_config(self):
        return {"params": self.params}

    def _push_weights(self, which=("g", "f", "h", "e")):
        ids = {"g": _lib.NET_G, "f": _lib.NET_F, "h": _lib.NET_H, "e": _lib.NET_E}
        for k in which:
            self.engine.set_weights(ids[k], self.nets[k])

    def set_weights(self, **nets):
        """Install network parameters ([(W, b), ...] per net) -- the counterpart of restoring a checkpoint."""
        for k, v in nets.items():
            self.nets[k] = [(np.asarray(W, np.float32), np.asarray(b, np.float32)) for W, b in v]
        self._push_weights(tuple(nets))

    def _next_seed(self):
        self._seed_counter += 1
        return (self._base_seed * 1000003 + self._seed_counter) & 0x7FFFFFFFFFFFFFFF

    def _dev(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.engine.device, dtype=torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.engine.device)

    def _panel_v_var(self, v):
        """mean per-column variance of this rank's rows of V (1 for a standardised panel), cached per tensor: diagnostics.py compares
        MSE_v / var(V)"""
        key = (v.data_ptr(), tuple(v.shape))
        if getattr(self, '_v_var_key', None) != key:
            self._v_var_key = key
            self._v_var = float(v.var(dim=0, unbiased=False).mean().item()) if v.shape[0] > 1 else 1.0
        return self._v_var

    # ------------------------------------------------------------------ fit
    def _net_dims(self, k):
        return [self.nets[k][0][0].shape[0]] + [W.shape[1] for W, _ in self.nets[k]]

    def _pull_weights(self, which=("g", "f", "h")):
        ids = {"g": _lib.NET_G, "f": _lib.NET_F, "h": _lib.NET_H, "e": _lib.NET_E}
        for k in which:
            self.nets[k] = self.engine.get_weights(ids[k], self._net_dims(k))

    @staticmethod
    def _choice_no_replace(n, k):
        """np.random.choice(n, k, replace=False) (base.py:406,413).  NumPy's implementation permutes all n
        indices per call (O(n)); for large panels draw k distinct indices by rejection instead (same law)."""
        if n <= 200000 or k * 20 > n:
            return np.random.choice(n, k, replace=False)
        while True:
            idx = np.random.randint(0, n, size=k)
            if len(np.unique(idx)) == k:
                return idx

    def egm_init(self, data, egm_n_iter=30000, batch_size=32, egm_batches_per_eval=500, verbose=1):
        """EGM warm start (base.py:380-431): g_d_freq WGAN-GP steps on the latent discriminator dz_net, then one
        step on g, e, f, h, per iteration.  Every step is one launch of the hand-written kernels of
        csrc/egm_kernels.h (forward, backward incl. the gradient-penalty double backward through the
        BatchNorm'd discriminator, Keras Adam).  The host draws the minibatch indices, the prior samples and the
        interpolation coefficients in the reference's order, one evaluation period at a time."""
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
        xd, yd, vd = self._dev(data_x[lo_r:hi_r]).reshape(-1), self._dev(data_y[lo_r:hi_r]).reshape(-1), self._dev(data_v[lo_r:hi_r])
        q = sum(p_["z_dims"])
        dims = [q] + list(p_["dz_units"]) + [1]
        dz = {"W": [_glorot(self._rs, dims[i], dims[i + 1]) for i in range(len(dims) - 1)],       # networks/base.py:338-363
              "b": [np.zeros(dims[i + 1], np.float32) for i in range(len(dims) - 1)],
              "gamma": [np.ones(dims[i + 1], np.float32) for i in range(len(dims) - 2)],
              "beta": [np.zeros(dims[i + 1], np.float32) for i in range(len(dims) - 2)]}
        self._push_weights()
        eng.egm_begin(b_loc, p_["dz_units"], p_["lr"], p_["use_z_rec"], dz)
        out_d = torch.zeros(2, device=dev)
        out_g = torch.zeros(6, device=dev)
        if world > 1:
            n_gen, n_dz = eng.egm_sizes()
            buf_g, buf_d = torch.empty(n_gen, device=dev), torch.empty(n_dz, device=dev)
        egm_log = []
        if verbose:
            print('EGM Initialization Starts ...')
        g_d_freq = int(p_['g_d_freq'])
        steps = g_d_freq + 1
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
                        eng.egm_disc_step(z_d[i, j], idx_d[i, j], vd, eps_h[i, j], out=out_d)
                    eng.egm_gen_step(z_d[i, g_d_freq], idx_d[i, g_d_freq], vd, xd, yd, out=out_g)
                for i in range(n_it if world > 1 else 0):
                    for j in range(g_d_freq):
                        eng.egm_disc_step(z_d[i, j], idx_d[i, j], vd, eps_h[i, j], apply=False, out=out_d)
                        eng.egm_grad(1, 1.0 / world, buf_d)
                        parallel.all_reduce_sum_(buf_d)                      # dz gradient of the global minibatch
                        eng.egm_apply(1, buf_d)
                    eng.egm_gen_step(z_d[i, g_d_freq], idx_d[i, g_d_freq], vd, xd, yd, apply=False, out=out_g)
                    eng.egm_grad(0, 1.0 / world, buf_g)
                    parallel.all_reduce_sum_(buf_g)                          # fused g | e | f | h gradient
                    eng.egm_apply(0, buf_g)
                batch_iter = stop
                if batch_iter % egm_batches_per_eval == 0:
                    eng.egm_sync()
                    self._pull_weights(("g", "f", "h", "e"))
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
            self._pull_weights(("g", "f", "h", "e"))
        # (the thresholds describe the full-length warm start: a short one is merely unconverged and gets no diagnosis)
        self._egm_late_l2z = diagnostics.late_l2_loss_z([a for a, _ in egm_log], [b for _, b in egm_log], egm_n_iter) if (egm_log and egm_n_iter >= diagnostics.MIN_EGM_ITER) else None
        self._second_optimum_warned = False
        if verbose:
            print('EGM Initialization Ends.')

    def fit(self, data, epochs=100, epochs_per_eval=5, batch_size=32, startoff=0, use_egm_init=True,
            egm_n_iter=30000, egm_batches_per_eval=500, save_format='txt', verbose=1, z_adam=None, host_loop=False, dp_comm=None):
        """Iterative theta / Z updates (base.py:434-532).

        ``host_loop=False`` (single process): the minibatches of an epoch are issued by ONE library call (bgm_causal_fit_epoch), the
        latent phase of a minibatch overlapping the theta phase of the next on a second stream; ``True`` keeps the per-minibatch
        calls from Python (same results bit for bit).  Under torch.distributed over RCCL (one GPU per rank) the epoch is ONE library call
        too (bgm_causal_fit_epoch_dp: the gradient all-reduce is enqueued from C++ between the gradient tiles and the Adam step); other
        process groups (gloo) keep the host loop, where `parallel.all_reduce_sum_` sits between the calls.

        ``batch_size`` is the GLOBAL minibatch; under torch.distributed every rank owns a contiguous row
        shard, draws its share of each minibatch from its own rows, and the g/f/h gradients are
        all-reduced (RCCL) before the Adam step, so all ranks hold identical networks.
        ``z_adam`` -- the latent optimizer (base.py:246-302 applies a sparse gradient with Keras' Adam, whose sparse path decays the
        moments of and updates ALL rows at every minibatch):
          "replay" (default) the same recursion with the zero-gradient steps of the rows outside a minibatch deferred until the row
                   is next used (O(batch) per step; equal to "dense" up to fp32 rounding of a 256-term series, csrc/z_replay.h);
          "dense"  the recursion as Keras executes it: a sweep over the [N x q] table per minibatch (bit-faithful order of operations);
          "lazy"   batch rows only -- a different optimizer (build option)."""
        if z_adam is None:
            z_adam = "replay"
            diagnostics.notice_once("z_adam", "fit(z_adam=...) not given: the latent Adam runs in its replayed form ('replay': equal to Keras' "
                                    "dense-decay sweep up to fp32 rounding); 'dense' executes the sweep as the reference does (DESIGN_HISTORY.md section 4d)")
        if use_egm_init:
            self.egm_init(data, egm_n_iter=egm_n_iter, batch_size=batch_size,
                          egm_batches_per_eval=egm_batches_per_eval, verbose=verbose)
        data_x, data_y, data_v = data
        n_total = len(data_x)
        if self._p['save_res'] and parallel.rank() == 0:
            with open('{}/params.txt'.format(self.save_dir), 'w') as f_params:
                f_params.write(str(self.params))
        q = self.engine.q
        lo_r, hi_r = parallel.shard_range(n_total)
        if use_egm_init:
            if verbose:
                print('Initialize latent variables Z with e(V)...')
            data_z_init = None                                                          # base.py:479
        else:
            if verbose:
                print('Random initialization of latent variables Z...')
            data_z_init = np.random.normal(0, 1, size=(n_total, q)).astype('float32')   # base.py:482
        n_loc = hi_r - lo_r
        world = parallel.world_size()
        dev = self.engine.device
        x = self._dev(data_x[lo_r:hi_r]).reshape(-1)
        y = self._dev(data_y[lo_r:hi_r]).reshape(-1)
        v = self._dev(data_v[lo_r:hi_r])
        self.data_z = self.engine.encode(v) if data_z_init is None else self._dev(data_z_init[lo_r:hi_r])
        zm = torch.zeros_like(self.data_z)
        zv = torch.zeros_like(self.data_z)
        b_loc = max(1, batch_size // world)
        # Every rank must take the same number of steps with the same batch sizes (one all-reduce per step): an epoch uses the
        # first n_total // world entries of each rank's permutation; a rank that owns one row more leaves one (random) row out.
        n_use = n_total // world if world > 1 else n_loc
        eng = self.engine
        n_params = eng.fit_begin(n_loc, b_loc)
        if self._restored_opt is not None:
            if len(self._restored_opt["m"]) == n_params:
                eng.fit_state(self._restored_opt)      # g / f / h_optimizer slots and step counters of the restored checkpoint
            else:
                import warnings
                warnings.warn("bayesgm_amd: the restored checkpoint's optimizer slots (%d values) do not match this model's %d "
                              "parameters and are discarded: the fit starts with fresh Adam state" % (len(self._restored_opt["m"]), n_params))
        self._restored_opt = None
        self._fit_live = (zm, zv)
        grad = torch.empty(n_params, device=dev, dtype=torch.float32)
        loss = torch.zeros(8, device=dev, dtype=torch.float64)       # theta phase: row sums of loss_v, |v-mu|^2, loss_x, ...
        loss_z = torch.zeros(8, device=dev, dtype=torch.float64)     # Z phase: [6] = row sums of the negative log joint
        if z_adam not in ("dense", "lazy", "replay"):
            raise ValueError("z_adam must be 'replay', 'dense' or 'lazy'")
        lazy = {"dense": 0, "lazy": 1, "replay": 2}[z_adam]
        replay = (lazy == 2)
        lr_z = self._p['lr_z']
        best_loss = np.inf
        # one process per GPU over RCCL: the epoch stays ONE library call, with an RCCL communicator of the library's own
        # (parallel.DeviceComm) for the per-step gradient all-reduce; other process groups (gloo) keep the per-minibatch host loop
        # (``dp_comm``: a parallel.DeviceComm to use instead -- a one-rank communicator runs the data-parallel call in a single process)
        if dp_comm is None and world > 1 and host_loop is False:
            dp_comm = parallel.fit_comm(dev)
        self.last_fit_path = ("library_epoch_dp (RCCL all-reduce inside bgm_causal_fit_epoch_dp, %d ranks)" % dp_comm.world if dp_comm is not None
                              else "library_epoch" if (world == 1 and host_loop is False) else "host_loop")
        # per-epoch trace (row means over the epoch's minibatches; the reference shows the last minibatch in its progress bar)
        self.fit_history = []
        if verbose:
            print('Iterative Updating Starts ...')
        try:
            for epoch in range(epochs + 1):
                # permutation of the LOCAL rows (np.random.choice(N, N, replace=False), base.py:489)
                sample_idx = torch.from_numpy(np.random.choice(n_loc, n_loc, replace=False).astype(np.int32)).to(dev)
                loss.zero_()
                loss_z.zero_()
                n_rows = 0
                if dp_comm is not None:                        # data parallel: the loop AND the all-reduce inside the library
                    eng.fit_epoch_dp(dp_comm, x, y, v, self.data_z, zm, zv, sample_idx[:n_use], b_loc, self._p['lr_theta'], lr_z, lazy, loss, loss_z)
                    n_rows = n_use
                elif world == 1 and host_loop is False:        # the minibatch loop inside the library (bgm_causal_fit_epoch)
                    eng.fit_epoch(x, y, v, self.data_z, zm, zv, sample_idx[:n_use], b_loc, self._p['lr_theta'], lr_z, lazy, loss, loss_z)
                    n_rows = n_use
                for i in (range(0, n_use, b_loc) if n_rows == 0 else ()):
                    idx = sample_idx[i:min(i + b_loc, n_use)]
                    bg = int(idx.numel()) * world
                    if replay:
                        eng.fit_z_sync(self.data_z, zm, zv, idx, lr_z)           # this minibatch's rows, current before they are read
                    eng.fit_theta_grad(x, y, v, self.data_z, idx, bg, grad, loss)
                    parallel.all_reduce_sum_(grad)                       # C1: fused g|f|h gradient
                    eng.fit_theta_apply(grad, self._p['lr_theta'])
                    eng.fit_z_step(x, y, v, self.data_z, zm, zv, idx, bg, self._p['lr_z'], lazy, loss_z)
                    n_rows += int(idx.numel())
                l = loss.cpu().numpy() / max(1, n_rows)
                lz = loss_z.cpu().numpy() / max(1, n_rows)
                self.fit_history.append(dict(epoch=epoch, loss_v=float(l[0]), loss_mse_v=float(l[1] / eng.v_dim), loss_x=float(l[2]),
                                             loss_mse_x=float(l[3]), loss_y=float(l[4]), loss_mse_y=float(l[5]),
                                             loss_postrior_z=float(lz[6])))
                if verbose:
                    print('Epoch [%d/%d]: loss_px_z [%.4f], loss_mse_x [%.4f], loss_py_z [%.4f], loss_mse_y [%.4f], loss_pv_z [%.4f], '
                          'loss_mse_v [%.4f], loss_postrior_z [%.4f]' % (epoch, epochs, l[2], l[3], l[4], l[5], l[0], l[1] / eng.v_dim, lz[6]))
                if epoch % epochs_per_eval == 0:
                    if replay:
                        eng.fit_z_sync(self.data_z, zm, zv, None, lr_z)          # flush: evaluate / checkpoints read the whole table
                    causal_pre, mse_x, mse_y, mse_v = self._evaluate_dev(x, y, v, self.data_z, n_total, lo_r)
                    self.fit_history[-1].update(mse_x=float(mse_x), mse_y=float(mse_y), mse_v=float(mse_v))
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
                            self._pull_weights()
                            self.save_checkpoint(epoch)
                    if self._p['save_res'] and parallel.rank() == 0:
                        save_data('{}/causal_pre_at_{}.{}'.format(self.save_dir, epoch, save_format), causal_pre)
        finally:
            self._fit_live = None
            if replay:
                eng.fit_z_sync(self.data_z, zm, zv, None, lr_z)
            eng.fit_end()
            self._pull_weights()

    def save_checkpoint(self, epoch):
        """ckpt_manager.save(epoch) (base.py:527-529).  The archive holds what the reference's tf.train.Checkpoint tracks
        (:112-122) -- the parameters of g, e, f, h, and, when written from inside `fit`, the Adam slots and step counters of the
        g / f / h optimizers and the latent optimizer's step counter; at most 5 are kept."""
        flat = {}
        for k, net in self.nets.items():
            for i, (W, b) in enumerate(net):
                flat["%s_W%d" % (k, i)] = W
                flat["%s_b%d" % (k, i)] = b
        if self._fit_live is not None:
            st = self.engine.fit_state()
            # (the latent table and its slots are not among the objects the reference's tf.train.Checkpoint tracks, :112-122: a
            # restored model re-initialises Z with e(V) or N(0, I) in fit, as the reference does -- they are not written)
            flat.update(opt_m=st["m"], opt_v=st["v"], opt_steps=np.array([st["t_theta"], st["t_z"]], np.int64))
        flat["seed_state"] = np.array([self._base_seed, self._seed_counter], np.int64)
        flat.update(self._checkpoint_extra())
        path = self.ckpt_manager.save("ckpt-%s.npz" % epoch, flat)
        print('Saving checkpoint for epoch {} at {}'.format(epoch, path))
        return path

    def _checkpoint_extra(self):
        """Further tracked objects of a subclass (IdentifiableCausalBGM: prior_net, prior_optimizer) as named arrays."""
        return {}

    def _restore_extra(self, d):
        pass

    def load_checkpoint(self, path):
        d = np.load(path)
        self._restore_extra(d)
        for k in list(self.nets):
            self.nets[k] = [(d["%s_W%d" % (k, i)], d["%s_b%d" % (k, i)]) for i in range(len(self.nets[k]))]
        self._push_weights()
        if "opt_m" in d.files:
            self._restored_opt = dict(m=d["opt_m"], v=d["opt_v"], t_theta=int(d["opt_steps"][0]), t_z=int(d["opt_steps"][1]))
        if "seed_state" in d.files and int(d["seed_state"][0]) == self._base_seed:
            self._seed_counter = int(d["seed_state"][1])

    # ------------------------------------------------------------------ evaluate
    @staticmethod
    def _percentile_nearest(a, qpct):
        """tfp.stats.percentile default interpolation='nearest' (base.py:558-559)."""
        s = np.sort(np.asarray(a).ravel())
        return s[int(np.round((len(s) - 1) * qpct / 100.0))]

    def _evaluate_dev(self, x, y, v, z, n_total, lo_r, nb_intervals=200, x_full=None):
        """evaluate on this rank's rows + reduction over ranks.  Returns numpy (causal_pre, mse_x, mse_y, mse_v)."""
        eng = self.engine
        n_loc = v.shape[0]
        if self._p['binary_treatment']:
            sums, ite = eng.evaluate(x, y, v, z)
            parallel.all_reduce_sum_(sums)
            ite = parallel.all_gather_rows(ite.reshape(-1, 1), n_total)
            s = sums.cpu().numpy()
            return (ite.cpu().numpy(), np.float32(s[1] / n_total), np.float32(s[2] / n_total),
                    np.float32(s[0] / (n_total * self.engine.v_dim)))
        # dose grid between the 5th / 95th percentile of ALL x (base.py:558-560)
        xs_all = parallel.all_gather_rows(x.reshape(-1, 1), n_total).cpu().numpy() if x_full is None else x_full
        x_min = self._percentile_nearest(xs_all, 5.0)
        x_max = self._percentile_nearest(xs_all, 95.0)
        x_values = np.linspace(x_min, x_max, nb_intervals).astype(np.float32)
        sums, dose = eng.evaluate(x, y, v, z, x_values)
        dose = dose.double()
        parallel.all_reduce_sum_(sums)
        parallel.all_reduce_sum_(dose)
        s = sums.cpu().numpy()
        return ((dose / n_total).float().cpu().numpy(), np.float32(s[1] / n_total), np.float32(s[2] / n_total),
                np.float32(s[0] / (n_total * self.engine.v_dim)))

    def evaluate(self, data, data_z=None, nb_intervals=200):
        """(causal_pre, mse_x, mse_y, mse_v) (base.py:534-570); data_z=None -> e_net(data_v)."""
        data_x, data_y, data_v = data
        n_total = len(data_x)
        lo_r, hi_r = parallel.shard_range(n_total)
        x = self._dev(data_x[lo_r:hi_r]).reshape(-1)
        y = self._dev(data_y[lo_r:hi_r]).reshape(-1)
        v = self._dev(data_v[lo_r:hi_r])
        if data_z is None:
            z = self.engine.encode(v)
        else:
            z = self._dev(data_z[lo_r:hi_r]) if len(data_z) == n_total else self._dev(data_z)
        return self._evaluate_dev(x, y, v, z, n_total, lo_r, nb_intervals)

    # ------------------------------------------------------------------ predict
    def predict(self, data, alpha=0.01, n_mcmc=3000, burn_in=5000, x_values=None, q_sd=1.0, sample_y=True,
                bs=10000, verbose=1):
        """Causal effects with posterior intervals from latent MCMC samples (base.py:573-668).

        ``bs`` bounded the host memory of the reference; here all rows are sampled in one launch per
        segment (row-blocked only if the ITE draw matrix would exceed device memory) and the result does
        not depend on it.  Under torch.distributed the rows are sharded by rank and results gathered."""
        assert 0 < alpha < 1, "The significance level 'alpha' must be greater than 0 and less than 1."
        parallel.check_n_mcmc(n_mcmc)
        binary = bool(self._p['binary_treatment'])
        if not binary and x_values is None:
            raise ValueError("For continuous treatment, 'x_values' must not be None. Provide a list or a single treatment value.")
        if x_values is not None:
            x_values = np.array([x_values], dtype=float) if np.isscalar(x_values) else np.array(x_values, dtype=float)
        data_x, data_y, data_v = data
        n_test = len(data_x)
        bs = max(1, int(bs))
        adaptive = (q_sd is None) or (q_sd <= 0)
        seed = self._next_seed()
        if verbose:
            print('MCMC Latent Variable Sampling ...')
        eng = self.engine
        dev = eng.device
        total_it = burn_in + n_mcmc
        world, rank = parallel.world_size(), parallel.rank()
        # Row blocks.  A fixed proposal scale makes every chain independent of its block (the Philox stream is keyed by the
        # global row), so each rank samples its contiguous shard in one piece.  With the adaptive scale (q_sd <= 0) the
        # reference adapts q_sd per `bs`-block from that block's acceptance window (base.py:632-637, 880-893): the blocks are
        # the reference's [start, start + bs) and are dealt to the ranks whole.
        if adaptive:
            blocks = [(s0, min(s0 + bs, n_test)) for s0 in range(0, n_test, bs)][rank::world]
        else:
            lo_r, hi_r = parallel.shard_range(n_test)
            blocks = [(lo_r, hi_r)] if hi_r > lo_r else []
        if binary:        # keep the [rows x n_mcmc] draw matrix of one launch below ~32 GiB
            max_rows = max(16, int((32 << 30) // (4 * max(1, n_mcmc))))
            if not adaptive:
                blocks = [(s0, min(s0 + max_rows, e0)) for (b0, e0) in blocks for s0 in range(b0, e0, max_rows)]
        acc_tail = 0.0
        if binary:
            res = torch.zeros((3, n_test), device=dev, dtype=torch.float32)     # mean, lower, upper (this rank's rows filled)
        else:
            sums = torch.zeros((len(x_values), n_mcmc), device=dev, dtype=torch.float64)   # adrf_draw_sums (base.py:660)
        for (s0, e0) in blocks:
            x = self._dev(data_x[s0:e0]).reshape(-1)
            y = self._dev(data_y[s0:e0]).reshape(-1)
            v = self._dev(data_v[s0:e0])
            if binary:
                out = eng.mh_sample(x, y, v, burn_in, n_mcmc, q_sd, seed, effect=_lib.EFFECT_ITE, sample_y=sample_y,
                                    row_base=s0, adaptive=adaptive)
                mean, lo, hi = eng.row_mean_quantiles(out["ite"], alpha / 2, 1 - alpha / 2)
                res[0, s0:e0], res[1, s0:e0], res[2, s0:e0] = mean, lo, hi
            else:
                out = eng.mh_sample(x, y, v, burn_in, n_mcmc, q_sd, seed, effect=_lib.EFFECT_ADRF, x_values=x_values,
                                    sample_y=sample_y, row_base=s0, adaptive=adaptive)
                sums += out["adrf"].double() * float(e0 - s0)
            acc_tail += float(out["acc_count"][max(0, total_it - 100):].sum().item())
        self._report_acceptance(acc_tail, min(100, total_it), n_test, verbose)
        if binary:
            parallel.all_reduce_sum_(res)                         # disjoint row sets: the sum is the gather
            res = res.cpu().numpy()
            return res[0], np.stack([res[1], res[2]], axis=1)
        parallel.all_reduce_sum_(sums)                            # C3: [n_doses x n_mcmc]
        causal_effects = (sums / float(n_test)).float().contiguous()
        adrf, lo, hi = eng.row_mean_quantiles(causal_effects, alpha / 2, 1 - alpha / 2)
        return adrf.cpu().numpy(), torch.stack([lo, hi], dim=1).cpu().numpy()

    def _report_acceptance(self, acc_tail, window, n_test, verbose):
        t = torch.tensor([acc_tail], dtype=torch.float64, device=self.engine.device)
        parallel.all_reduce_sum_(t)
        self.last_acceptance_rate = float(t.item()) / (window * n_test)
        if verbose:
            print(f"Final MCMC Acceptance Rate: {self.last_acceptance_rate:.4f}")

    def metropolis_hastings_sampler(self, data, initial_q_sd=1.0, q_sd=None, burn_in=5000, n_keep=3000,
                                    target_acceptance_rate=0.25, tolerance=0.05, adjustment_interval=50,
                                    adaptive_sd=None, window_size=100):
        """Posterior samples of Z, shape (n_keep, n, q) (base.py:820-904)."""
        data_x, data_y, data_v = data
        if adaptive_sd is None:
            adaptive_sd = (q_sd is None or q_sd <= 0)
        out = self.engine.mh_sample(self._dev(data_x).reshape(-1), self._dev(data_y).reshape(-1), self._dev(data_v),
                                    burn_in, n_keep, q_sd, self._next_seed(), want_draws=True, adaptive=adaptive_sd,
                                    initial_q_sd=initial_q_sd, target=target_acceptance_rate, tol=tolerance,
                                    adj_int=adjustment_interval, window=window_size)
        tot = burn_in + n_keep
        w = min(window_size, tot)
        self.last_acceptance_rate = float(out["acc_count"][tot - w:].sum().item()) / (w * len(data_x))
        print(f"Final MCMC Acceptance Rate: {self.last_acceptance_rate:.4f}")
        return out["draws"].cpu().numpy()

    def infer_from_latent_posterior(self, data_posterior_z, x_values=None, sample_y=True, eps=1e-6, seed=None):
        """Causal effects from posterior draws of Z, shape (n_keep, n, q) (base.py:671-763): binary treatment -> ITE draws
        (n_keep, n); continuous -> ADRF draws (len(x_values), n_keep).  `predict` computes the same quantities inside the
        sampling kernel without materialising the draws; this is the stand-alone form of the reference."""
        draws = self._dev(data_posterior_z)
        if not self._p["binary_treatment"] and x_values is None:
            raise ValueError("For continuous treatment, `x_values` must not be None. Provide a list or numpy array.")
        n = draws.shape[1]
        x0 = torch.zeros(n, device=self.engine.device)          # the treatment slot is overwritten by the counterfactual dose
        out = self.engine.effects(x0, draws, 0, self._next_seed() if seed is None else seed, x_values=x_values, sample_y=sample_y)
        return out.cpu().numpy()

    def get_log_posterior(self, data_x, data_y, data_v, data_z, eps=1e-6):
        """log p(z | x, y, v) + const, shape (n,) (base.py:765-817)."""
        return self.engine.logpost(self._dev(data_x).reshape(-1), self._dev(data_y).reshape(-1), self._dev(data_v),
                                   self._dev(data_z)).cpu().numpy()
