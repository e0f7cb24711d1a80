/*
 * bgm_hip.h -- C ABI of the MI355X (gfx950) hot-path library  libbgm_hip.so
 *
 * The reference (liuq-lab/bayesgm v1.0.2) is pure Python on TensorFlow: it has
 * NO plugin / FFI interface.  Each entry point below therefore replaces one
 * @tf.function (or the NumPy loop around it) of the reference; the citation
 * after "replaces:" is the reference file:line whose arithmetic it computes.
 * The reference-side binding a maintainer would add is a ctypes stub -- see
 * INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no torch / C++ types in any signature;
 *   - every `*_dev` pointer is a DEVICE pointer owned by the caller
 *     (row-major float32 unless stated), every `*_host` pointer is host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls
 *     are asynchronous on that stream unless stated;
 *   - return value: 0 = ok, negative = error (BGM_E_*), message via
 *     bgm_last_error() (thread-local);
 *   - a handle is bound to one device and is thread-compatible (one caller at a
 *     time per handle).
 */
#ifndef BGM_HIP_H
#define BGM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the entry points below are its only exported symbols. */
#define BGM_API __attribute__((visibility("default")))

#define BGM_OK 0
#define BGM_E_INVALID (-1)     /* bad argument / unsupported shape            */
#define BGM_E_HIP (-2)         /* HIP runtime error                           */
#define BGM_E_STATE (-3)       /* call order (weights not set, ...)           */
#define BGM_E_UNSUPPORTED (-4) /* shape outside the compiled kernel variants  */

#define BGM_MAX_LAYERS 8

#define BGM_NET_G 0 /* z -> (mu_v[p], s_v)           causalbgm/base.py:65,74  */
#define BGM_NET_F 1 /* (z0,z1,x) -> (mu_y, s_y)      causalbgm/base.py:69,78  */
#define BGM_NET_H 2 /* (z0,z2) -> (mu_x|logit, s_x)  causalbgm/base.py:71,80  */
#define BGM_NET_E 3 /* v -> z  (encoder)             causalbgm/base.py:67,76  */

#define BGM_EFFECT_NONE 0
#define BGM_EFFECT_ADRF 1 /* continuous treatment: dose-response sums          */
#define BGM_EFFECT_ITE 2  /* binary treatment: individual treatment effects    */

typedef struct bgm_handle bgm_handle;

/* Shape of a CausalBGM model (the `params` dict of causalbgm/base.py:56-93). */
typedef struct {
  int32_t v_dim;            /* p                                              */
  int32_t z_dims[4];        /* [z0, z1, z2, z3]; q = sum                      */
  int32_t binary_treatment; /* 0 continuous (Gaussian x), 1 binary (BCE)      */
  int32_t n_hidden_g, g_units[BGM_MAX_LAYERS];
  int32_t n_hidden_f, f_units[BGM_MAX_LAYERS];
  int32_t n_hidden_h, h_units[BGM_MAX_LAYERS];
  int32_t n_hidden_e, e_units[BGM_MAX_LAYERS];
  float sigma_v, sigma_x, sigma_y; /* params['sigma_*'] if > 0, else learned
                                      variance head softplus(.)+1e-6           */
} bgm_causal_config;

BGM_API const char *bgm_last_error(void);
BGM_API const char *bgm_version(void);

/* Create / destroy a per-device handle.  Synchronous. */
BGM_API int bgm_create(bgm_handle **out, int device);
/* BatchNormalization mode of the Discriminator networks (networks/base.py:338-385) of every EGM warm-start session opened
 * afterwards on this handle (bgm_causal_egm_begin, bgm_bnn_egm_begin, bgm_bgm_egm_begin, bgm_bvn_egm_begin):
 *   0 (default of the library)  batch statistics -- `norm_layer(x)` inside a Model.call(training=True) under Keras' rule that an
 *                               inner layer inherits the outer call's training mode;
 *   1                           inference mode on the layer's initial moving averages (mean 0, variance 1), the behaviour the
 *                               reference's published training log is consistent with (DESIGN_HISTORY.md section 2b). */
BGM_API int bgm_set_disc_norm(bgm_handle *h, int32_t mode);
/* Arithmetic of the CausalBGM sampling kernels launched afterwards through bgm_causal_logpost / bgm_causal_mh_run:
 *   0 (default)  fp32 MFMA -- the reference's arithmetic (causalbgm/base.py:765-904 run in float32);
 *   1            split precision "bf16 x 3": weights and activations as sums of two bf16 numbers, three bf16 MFMA products per
 *                contraction with fp32 accumulation (relative error ~6e-6 per layer against 2.4e-7 in fp32).  Same algorithm,
 *                RNG streams and outputs; chains agree with the fp32 ones statistically, not draw for draw (DESIGN_HISTORY.md section 4b);
 *   2            the same kernels on fp16 operands ("f16 x 3": hi + lo carry 22 mantissa bits, the log posterior is within the
 *                fp32 kernel's own distance of float64; weights beyond 65504 are clamped, an activation beyond 65504 overflows: fp16 range).
 * Modes 1 and 2 exist for the LDS-resident (default-width) shapes; mode 2 also outside them for hidden widths up to 128 (the general-width
 * engine's row-tile-per-wave kernels: fp32 activations in LDS split into hi / lo fp16 at the operand load, csrc/gx_device.h gx_dense_x3).
 * Other shapes answer BGM_E_UNSUPPORTED at the sampling call. */
BGM_API int bgm_causal_set_precision(bgm_handle *h, int32_t mode);
/* Conditional latent prior Z | U ~ N(mu(U), sigma^2(U) I) of IdentifiableCausalBGM (models/causalbgm/identifiable.py:195-211,
 * 541-551) for the sampling calls made afterwards (bgm_causal_logpost, bgm_causal_mh_run; fp32 and split-precision kernels): seg_dev [n] = segment of
 * every LOCAL row of those calls (int32), tab_dev [n_segments x (q + 2)] = per segment mu [q], 1 / sigma^2, (q / 2) log sigma^2.
 * Both NULL: back to the standard-normal prior.  The buffers must stay valid while set. */
BGM_API int bgm_causal_set_prior(bgm_handle *h, const int32_t *seg_dev, const float *tab_dev, int32_t n_segments);

/* The prior network of IdentifiableCausalBGM, prior_net = BaseFullyConnectedNet(n_segments -> prior_units -> q + 1)
 * (identifiable.py:76-78; LeakyReLU(0.2) hidden layers, linear output).  Parameters, gradients and Adam slots live in caller-owned
 * device arrays in Keras order (W_0 [in x out] row-major, b_0, W_1, b_1, ...); bgm_prior_n_params gives their length. */
typedef struct {
  int32_t n_layers;          /* dense layers, 1..4 */
  int32_t dims[5];           /* n_segments, prior_units..., q + 1 */
} bgm_prior_config;
BGM_API int bgm_prior_n_params(const bgm_prior_config *cfg, int64_t *count);
/* table_dev [n_segments x (q + 2)] = per segment mu [q], 1 / sigma^2, (q / 2) log sigma^2 with sigma^2 = softplus(out[q]) + 1e-6
 * (identifiable.py:541-551): the tab_dev of bgm_causal_set_prior. */
BGM_API int bgm_prior_table(bgm_handle *h, const bgm_prior_config *cfg, const float *theta_dev, float *table_dev, void *stream);
/* replaces: the conditional-prior half of update_latent_variable_sgd, identifiable.py:195-226.  dz_dev [batch x q] is the output of
 * bgm_causal_fit_z_grad on the same rows (gradient of the batch-mean negative log joint with the standard-normal prior); the call
 * exchanges the prior term (- z / B + (z - mu(u)) / (sigma^2(u) B)), applies the latent step with FRESH Adam slots and step count
 * t_z (the batch latents are a new Variable in every minibatch, :304), and takes one Adam step (slots m_dev / v_dev, step count
 * t_prior, learning rate lr_prior) on the prior net with the gradient of the batch-mean prior term.  seg_dev [n_rows]: segment of
 * every row.  out_dev [2] (or NULL) = batch means of the conditional-prior term and of |z|^2 / 2. */
BGM_API int bgm_prior_step(bgm_handle *h, const bgm_prior_config *cfg, float *theta_dev, float *m_dev, float *v_dev, const int32_t *seg_dev,
                   float *data_z_dev, const int32_t *idx_dev, int32_t batch, const float *dz_dev, float lr_z, float lr_prior,
                   int64_t t_z, int64_t t_prior, float *out_dev, void *stream);
/* Data-parallel form of bgm_prior_step (one process per GPU, rows sharded): bgm_prior_grad takes the latent step on the rank's `batch`
 * rows (all batch means over batch_global = the rows of all ranks; dz_dev from bgm_causal_fit_z_grad with the same batch_global) and
 * leaves the prior net's gradient in grad_dev [bgm_prior_n_params] -> [caller: RCCL all-reduce(SUM)] -> bgm_prior_apply takes the Adam
 * step bgm_prior_step would have taken.  out_dev [2]: this rank's share of the two batch means (they add up over ranks). */
BGM_API int bgm_prior_grad(bgm_handle *h, const bgm_prior_config *cfg, const float *theta_dev, const int32_t *seg_dev, float *data_z_dev,
                   const int32_t *idx_dev, int32_t batch, int32_t batch_global, const float *dz_dev, float lr_z, int64_t t_z,
                   float *grad_dev, float *out_dev, void *stream);
BGM_API int bgm_prior_apply(bgm_handle *h, const bgm_prior_config *cfg, float *theta_dev, float *m_dev, float *v_dev, const float *grad_dev,
                    float lr_prior, int64_t t_prior, void *stream);
/* ---- the same with a BAYESIAN prior network, IdentifiableCausalBGM(use_bnn=True): prior_net = BayesianFullyConnectedNet(n_segments ->
 * prior_units -> q + 1) (identifiable.py:66-67; networks/bnn.py:4-38).  Parameter layout (as the nets of bgm_bnn_begin): gamma [k],
 * beta [k], then per layer loc [in x out], rho [in x out], bias [out]; bgm_bprior_n_params counts them.  norm_mode: input
 * BatchNormalization on 0 = the statistics of the batch at hand, 1 = fixed mean 0 / variance 1 (bgm_bnn_config.norm_mode).
 * Noise: key = seed, call id = stream_id, net id 4 in the streams of oracle/bnn.py; row b of the minibatch draws its sign words as row
 * row0 + b (row0: this rank's first position in the global minibatch).
 * bgm_bprior_step replaces the conditional-prior half of update_latent_variable_sgd with use_bnn (identifiable.py:195-226):
 * dz_dev [batch x q] = gradient of the batch-mean negative log joint with the standard-normal prior (bgm_bnn_z_step with dz_out_dev);
 * the call exchanges the prior term, takes the latent step with fresh Adam slots (step count t_z) and one Adam step (slots m_dev /
 * v_dev, step count t_prior, lr_prior) on the prior net with the gradient of  batch-mean prior term + kl_weight * sum KL(N(loc,
 * sigma^2) || N(0, 1)).  apply = 0 (data parallel): the latent step is taken, the DATA part of the net's gradient goes to grad_dev
 * -> [caller: RCCL all-reduce(SUM)] -> bgm_bprior_apply adds the KL part and takes the Adam step.  out_dev [3] (or NULL): batch
 * means of the conditional-prior term and of |z|^2 / 2 (this rank's share), sum of the KL terms. */
BGM_API int bgm_bprior_n_params(const bgm_prior_config *cfg, int64_t *count);
BGM_API int bgm_bprior_step(bgm_handle *h, const bgm_prior_config *cfg, int32_t norm_mode, float kl_weight, float *theta_dev, float *m_dev,
                    float *v_dev, const int32_t *seg_dev, float *data_z_dev, const int32_t *idx_dev, int32_t batch, int32_t batch_global,
                    int32_t row0, const float *dz_dev, float lr_z, float lr_prior, int64_t t_z, int64_t t_prior, uint64_t seed,
                    uint32_t stream_id, float *grad_dev, int32_t apply, float *out_dev, void *stream);
BGM_API int bgm_bprior_apply(bgm_handle *h, const bgm_prior_config *cfg, float kl_weight, float *theta_dev, float *m_dev, float *v_dev,
                     const float *grad_dev, float lr_prior, int64_t t_prior, void *stream);
/* Conditional prior of the SAMPLING calls of a Bayesian-network session (bgm_bnn_logpost, bgm_bnn_mh_run) made afterwards:
 * replaces prior_net(data_u) inside get_log_posterior (identifiable.py:541-551) -- one noisy call of the prior net per log-posterior
 * evaluation and block of rows (call ids as g, h, f: stream_id; 2 it and 2 it + 1 in the sampler), seg_dev [n] the segments of the
 * rows of those calls.  theta_dev = NULL clears it.  The prior net's input BatchNormalization follows the session's norm_mode:
 * 1 inference mode; 0 the statistics of the block of rows of the call (shares of the block's rows per segment, counted once per
 * (n, block_rows) after this call: the contents of seg_dev must not change until the next bgm_bnn_set_prior; at most 64 segments;
 * no shares of one block, block_row0 = 0). */
BGM_API int bgm_bnn_set_prior(bgm_handle *h, const bgm_prior_config *cfg, const float *theta_dev, const int32_t *seg_dev);
/* Arithmetic of the SAMPLING calls of a Bayesian-network session made afterwards (bgm_bnn_logpost, bgm_bnn_mh_run, bgm_bnn_effects --
 * get_log_posterior / metropolis_hastings_sampler / infer_from_latent_posterior of causalbgm/base.py:671-904 on the Flipout nets of
 * networks/bnn.py:4-38):
 *   0 (default)  fp32 MFMA;
 *   2            split precision "f16 x 3": posterior means, perturbations and activations as sums of two fp16 numbers (22 mantissa
 *                bits), three fp16 MFMA products per contraction with fp32 accumulation, sign flips as XORs on the packed words
 *                (csrc/bnx_kernels.h).  Same algorithm, Philox streams, sign words and perturbation draws; the log posterior stays
 *                within the fp32 kernels' own distance of float64.  fp16 range: a weight beyond 65504 is clamped, an activation beyond it overflows.
 * Built for the sessions the default-shape kernels serve (inference-mode input normalisation, default widths); BGM_E_UNSUPPORTED otherwise
 * and for mode 1 (there is no bf16 form of this family).  The minibatch steps and bgm_bnn_evaluate stay fp32. */
BGM_API int bgm_bnn_set_precision(bgm_handle *h, int32_t mode);
BGM_API int bgm_destroy(bgm_handle *h);

/* Declare the model shape.  Synchronous.  replaces: CausalBGM.__init__ network
 * construction, causalbgm/base.py:64-84. */
/* Any hidden widths in [1, 4096] and up to 8 hidden layers per net are accepted (networks/base.py:4-51 takes any nb_units).  Two kernel
 * families serve them:
 *   - the reference's defaults (g_units = e_units = [64]*k, f_units = h_units = [64,32,8]): LDS-resident kernels when sum(z_dims) <= 19
 *     and v_dim <= 207 (<= 159 when sum(z_dims) > 11) -- the model runs on the smallest compiled kernel shape that contains it (zero
 *     padding, identical results); the general path of the same kernels for sum(z_dims) <= 31 at any v_dim;
 *   - everything else: the general-width engine (32-row activation tiles in LDS, padded weight packs streamed from L2; csrc/gx_api.hip).
 *     Its limit is the LDS tile: a call whose widest layer does not fit (hidden widths beyond ~550) returns BGM_E_UNSUPPORTED and says so.
 * Same entry points, RNG streams and results either way. */
BGM_API int bgm_causal_configure(bgm_handle *h, const bgm_causal_config *cfg);

/* Upload one network's parameters from HOST memory, flat float32 in Keras
 * order: for each Dense layer  W [in x out] row-major, then b [out]
 * (networks/base.py:17-26).  `count` = number of floats.  Synchronous on
 * `stream`.  Packs the weights into the MFMA fragment order used by the
 * kernels. */
BGM_API int bgm_causal_set_weights(bgm_handle *h, int net_id, const float *theta_host, int64_t count,
                           void *stream);

/* log p(z | x, y, v) up to a constant for n rows.
 * replaces: CausalBGM.get_log_posterior, causalbgm/base.py:765-817.
 * x,y [n], v [n x p], z [n x q] -> out [n]. */
BGM_API int bgm_causal_logpost(bgm_handle *h, const float *x_dev, const float *y_dev, const float *v_dev,
                       const float *z_dev, int64_t n, float *out_dev, void *stream);

/* Encoder forward  z = e(v)  for n rows (Z initialisation of fit and
 * evaluate(data_z=None)).  replaces: self.e_net(data_v), causalbgm/base.py:479,538. */
BGM_API int bgm_causal_encode(bgm_handle *h, const float *v_dev, int64_t n, float *z_dev, void *stream);

/* Arguments of one segment of the random-walk Metropolis-Hastings sampler.
 * replaces: CausalBGM.metropolis_hastings_sampler loop body, causalbgm/base.py:860-898,
 * fused with CausalBGM.infer_from_latent_posterior, :671-763, for retained draws. */
typedef struct {
  const float *x_dev, *y_dev, *v_dev; /* [n], [n], [n x p]                    */
  int64_t n;                          /* rows (independent chains)            */
  int64_t row_base;                   /* global index of row 0 (RNG counter)  */
  float *state_dev;                   /* [n x q] chain state, in/out          */
  float *logp_dev;                    /* [n] cached log posterior, in/out     */
  int32_t init;                       /* 1: draw state ~ N(0,1) (base.py:842)
                                         and compute logp before iterating    */
  int32_t it_begin, n_iters;          /* iterations [it_begin, it_begin+n)    */
  int32_t burn_in;                    /* draws are retained for it >= burn_in */
  float q_sd;                         /* proposal std-dev (base.py:862)       */
  uint64_t seed;                      /* Philox key                           */
  uint32_t *acc_count_dev;            /* [>= it_begin+n_iters] accepted-chain
                                         count per iteration (+=), or NULL    */
  float *draws_dev;                   /* [n_keep x n x q] retained states
                                         (base.py:896), or NULL               */
  int32_t n_keep;                     /* leading dim of draws / effects       */
  int32_t effect;                     /* BGM_EFFECT_*                         */
  int32_t sample_y;                   /* base.py:703,752                      */
  const float *x_values_dev;          /* [n_doses] (ADRF)                     */
  int32_t n_doses;
  float *adrf_partial_dev;            /* ADRF: [n_slots x n_keep x n_doses]
                                         per-wave-slot sums over rows (+=);
                                         n_slots from bgm_causal_mh_slots()   */
  float *ite_dev;                     /* ITE: [n x n_keep] draws, row-major
                                         per observation                      */
  uint64_t *clock_dev;                /* optional measurement aid: [n_slots][4] =
                                         shader cycles, 100 MHz ticks, start
                                         tick, XCC id of every wave; or NULL  */
} bgm_mh_args;

/* Number of wave slots (leading dim of adrf_partial) the MH kernel uses for n rows. */
BGM_API int bgm_causal_mh_slots(bgm_handle *h, int64_t n, int32_t *n_slots);

BGM_API int bgm_causal_mh_run(bgm_handle *h, const bgm_mh_args *args, void *stream);

/* out[k][d] = sum_s partial[s][d][k] / n_total  (fixed order => deterministic; the slots are draw-major -- a retained draw's doses
 * share a cache line in the sampling kernels' accumulation -- the result has the reference's [n_doses x n_keep] orientation).
 * replaces: adrf_draw_sums / n_seen, causalbgm/base.py:660-663. */
BGM_API int bgm_adrf_reduce(bgm_handle *h, const float *partial_dev, int32_t n_slots, int32_t n_doses,
                    int32_t n_keep, double n_total, float *out_dev, void *stream);

/* Per-row mean and linear-interpolated quantiles over m contiguous values:
 * in [n_rows x m] -> mean [n_rows], lo [n_rows], hi [n_rows].
 * replaces: np.mean / np.quantile(..., axis) at causalbgm/base.py:640-642,664-666.  Any m: rows of up to 32768 values are sorted
 * in LDS, longer rows have their order statistics selected by radix passes over memory (same results). */
BGM_API int bgm_row_mean_quantiles(bgm_handle *h, const float *in_dev, int64_t n_rows, int32_t m,
                           double q_lo, double q_hi, float *mean_dev, float *lo_dev,
                           float *hi_dev, void *stream);

/* Outcome-net cache of the fused effect samplers (bgm_causal_mh_run with BGM_EFFECT_ADRF / BGM_EFFECT_ITE on the LDS-resident kernels).
 * infer_from_latent_posterior (causalbgm/base.py:671-763) evaluates f(z, x_e) for every retained draw; consecutive draws of a
 * Metropolis-Hastings chain are equal whenever the proposal was rejected (base.py:868-871), and f is deterministic, so its (mean, sd)
 * at every dose are unchanged.  With the cache on (default), a retained iteration in which none of the 16 chains of a wave moved takes
 * them from the previous evaluation and only draws the new outcome noise: the ADRF sums / treatment effects are bit-identical, the
 * dose evaluations are skipped.  on = 0 evaluates the outcome net at every retained iteration, as the reference does.
 * bgm_causal_outcome_cache_stats: out2[0] = retained tile-iterations served from the cache, out2[1] = retained tile-iterations. */
/* on = 2 (the default since round 5): the cache works per CHAIN where the event form of the retained phase exists (dose-response sums on
 * the fp32 LDS-resident kernels, standard-normal prior, up to 32 doses; csrc/causal_event_kernels.h): the retained iterations run as the
 * pure-transition kernel, every accepted move appends an event (chain, iteration, state), the outcome net runs on dense 16-event tiles
 * and a third pass adds mean + sd * noise for every (row, retained draw) with the fused kernel's Philox calls, reductions and slot
 * order -- the sums are bit-identical to on = 0 / 1, only a fraction `acceptance rate` of the outcome-net evaluations remains.
 * Everywhere else on = 2 behaves as on = 1.  The event buffers of one segment of retained iterations are sized for the worst case
 * (every chain moves at every iteration); bgm_causal_set_event_budget bounds them (bytes; 0 = BGM_EVENT_BUDGET_MB or 8 GiB): the
 * retained phase is cut into segments that fit.  Stats in that form: out2[0] = retained chain-iterations that needed no evaluation,
 * out2[1] = retained chain-iterations. */
BGM_API int bgm_causal_set_outcome_cache(bgm_handle *h, int32_t on);
BGM_API int bgm_causal_set_event_budget(bgm_handle *h, int64_t bytes);
BGM_API int bgm_causal_outcome_cache_stats(bgm_handle *h, int64_t *out2, int32_t reset);

/* Kernel duration bookkeeping for bench.py: when enabled, bgm_causal_mh_run
 * brackets each kernel launch with hipEvents on the launch stream and
 * accumulates (launches, milliseconds) per kernel kind = BGM_EFFECT_* of the
 * launch (BGM_EFFECT_NONE = the pure-transition kernel); kind -1 = all.
 * Reading synchronises the pending events. */
BGM_API int bgm_timing_enable(bgm_handle *h, int enable);
BGM_API int bgm_timing_read(bgm_handle *h, int kind, int64_t *n_launches, double *total_ms, int reset);

/* Static facts of the selected MH kernel variant (for roofline accounting). */
typedef struct {
  int32_t rows_per_wave, waves_per_block, grid_blocks;
  int32_t mfma_per_transition_per_wave; /* issued 16x16x4 MFMAs                */
  int32_t lds_bytes;
  double flop_per_row_transition;       /* algorithmic 2*MACs(g+f+h)           */
} bgm_mh_info;
BGM_API int bgm_causal_mh_info(bgm_handle *h, int64_t n, bgm_mh_info *info);

/* Which kernels this handle runs, as text (diagnostics: bench.py prints it per rank): the sampling path, and -- between
 * bgm_causal_fit_begin and bgm_causal_fit_end -- the minibatch-step path for a local minibatch of `batch` rows (under data
 * parallelism batch = batch_size / world).  out: cap bytes, NUL-terminated. */
BGM_API int bgm_causal_describe(bgm_handle *h, int32_t batch, char *out, int32_t cap);

/* evaluate at a given latent matrix z [n x q]: accumulates sums[0..2] += {sum |v - mu_v|^2,
 * sum (x - x_pred)^2, sum (y - mu_y)^2} (divide by n*p, n, n for the MSEs) and the plug-in causal
 * estimate: binary -> ite_dev[n] = f(z0,z1,1) - f(z0,z1,0); continuous -> adrf_partial_dev
 * [n_slots x n_doses] += per-wave sums over rows of f(z0,z1,x_k) (reduce with bgm_adrf_reduce,
 * n_keep = 1; n_slots from bgm_causal_evaluate_slots).
 * replaces: CausalBGM.evaluate, causalbgm/base.py:534-570 (the dose grid / percentiles stay on the host). */
BGM_API int bgm_causal_evaluate(bgm_handle *h, const float *x_dev, const float *y_dev, const float *v_dev,
                        const float *z_dev, int64_t n, const float *x_values_dev, int32_t n_doses,
                        double *sums_dev, float *adrf_partial_dev, float *ite_dev, void *stream);
BGM_API int bgm_causal_evaluate_slots(bgm_handle *h, int64_t n, int32_t *n_slots);

/* replaces: CausalBGM.infer_from_latent_posterior, causalbgm/base.py:671-763, on a given tensor of posterior draws
 * draws_dev [n_keep x n x q] (the pass fused into bgm_causal_mh_run computes the same numbers from the same draws:
 * outcome noise of draw d = Philox(row_base + row, burn_in + d, dose block)).  binary -> ite_dev [n x n_keep];
 * continuous -> adrf_partial_dev [n_slots x n_keep x n_doses] (+=; n_slots = bgm_causal_evaluate_slots; reduce with
 * bgm_adrf_reduce). */
BGM_API int bgm_causal_effects(bgm_handle *h, const float *x_dev, const float *draws_dev, int64_t n, int64_t row_base,
                       int32_t n_keep, int32_t burn_in, uint64_t seed, int32_t sample_y, const float *x_values_dev,
                       int32_t n_doses, float *adrf_partial_dev, float *ite_dev, void *stream);

/* ------------------------------------------------------------------------------------------
 * CausalBGM.fit step functions (iterative theta / Z updates).
 * replaces: update_g_net :156-180, update_h_net :183-214, update_f_net :217-243,
 *           update_latent_variable_sgd :246-302 of causalbgm/base.py, and the Keras Adam
 *           optimizers created at :90-93 (beta_1 = 0.9, beta_2 = 0.99, epsilon = 1e-7).
 * One minibatch of the loop :494-505 is
 *     bgm_causal_fit_theta_grad  -> [caller: RCCL all-reduce(SUM) of grad_dev across ranks]
 *     bgm_causal_fit_theta_apply -> bgm_causal_fit_z_step
 * Losses are batch means over `batch_global` rows (the sum of the ranks' local batches), so the
 * all-reduced gradient is the gradient of the global-batch mean.
 * ------------------------------------------------------------------------------------------ */

/* Start a fit session: uploads the current g/f/h parameters to the device, zeroes the Adam slots
 * and step counters, sizes the workspace for minibatches of up to max_batch LOCAL rows out of
 * n_rows local rows.  Synchronous. */
BGM_API int bgm_causal_fit_begin(bgm_handle *h, int64_t n_rows, int32_t max_batch, void *stream);

/* Number of trainable parameters of g, f, h (= length of grad_dev). */
BGM_API int bgm_causal_fit_n_params(bgm_handle *h, int64_t *n_params);

/* Gradients of the three batch-mean losses w.r.t. theta_g | theta_f | theta_h (Keras order) for the
 * local minibatch rows idx_dev[0..batch) (int32 row indices into x/y/v/data_z; NULL = rows
 * row_lo .. row_lo+batch-1).  grad_dev [n_params] is overwritten.  loss_dev (double[8], may be NULL)
 * is incremented by {sum loss_v, sum |v-mu|^2, sum loss_x, sum (x-mu_x)^2 | sum bce, sum loss_y,
 * sum (y-mu_y)^2, unused, unused} over the local rows. */
BGM_API int bgm_causal_fit_theta_grad(bgm_handle *h, const float *x_dev, const float *y_dev, const float *v_dev,
                              const float *data_z_dev, const int32_t *idx_dev, int64_t row_lo,
                              int32_t batch, int32_t batch_global, float *grad_dev, double *loss_dev,
                              void *stream);

/* One Adam step on theta with the (all-reduced) gradient; refreshes the packed weights used by
 * every other kernel of the handle. */
BGM_API int bgm_causal_fit_theta_apply(bgm_handle *h, const float *grad_dev, float lr_theta, void *stream);

/* update_latent_variable_sgd for the local minibatch with the CURRENT networks: gradient of the
 * batch-mean negative log joint w.r.t. the batch rows of data_z and one Adam step on the latent
 * matrix.  zm/zv [n_rows x q] are the Adam slots.  lazy = 0 reproduces Keras' sparse-gradient Adam
 * (moment decay and update applied to ALL n_rows rows every step), lazy = 1 touches the batch rows
 * only (a different optimizer), lazy = 2 is lazy = 0 with the untouched rows' steps deferred (bgm_causal_fit_z_sync).  loss_dev[6] += sum over local rows of the per-row negative log joint. */
BGM_API int bgm_causal_fit_z_step(bgm_handle *h, const float *x_dev, const float *y_dev, const float *v_dev,
                          float *data_z_dev, float *zm_dev, float *zv_dev, const int32_t *idx_dev,
                          int64_t row_lo, int32_t batch, int32_t batch_global, float lr_z, int32_t lazy,
                          double *loss_dev, void *stream);

/* Replay mode of the latent optimizer (lazy = 2 in bgm_causal_fit_z_step): the same dense-decay Adam as lazy = 0, with the
 * zero-gradient steps of the rows outside a minibatch deferred until the row is next used.  For such a row a step is m <- b1 m,
 * v <- b2 v, z <- z - lr_t m / (sqrt(v) + eps): k pending steps are two powers and a geometric-like series of which 256 terms are
 * summed (the rest is < 1e-11 of the first), so a minibatch costs O(batch) instead of a sweep over the [n_rows x q] table, and the
 * table differs from the lazy = 0 one by fp32 rounding of the series only (tests/test_gpu_fit.py states the bound).
 *   idx_dev != NULL: bring the `batch` listed rows up to date.  Call it on a minibatch's rows BEFORE bgm_causal_fit_theta_grad
 *                    (which reads them) and bgm_causal_fit_z_step(lazy = 2), which refuses to run otherwise.
 *   idx_dev == NULL: flush -- bring every row up to date (before evaluate / predict / a checkpoint / reading data_z, and before
 *                    switching to another lazy mode).
 * lr_z is the (constant) learning rate of the latent optimizer (causalbgm/base.py:93). */
BGM_API int bgm_causal_fit_z_sync(bgm_handle *h, float *data_z_dev, float *zm_dev, float *zv_dev, const int32_t *idx_dev, int32_t batch,
                          float lr_z, void *stream);

/* A whole list of minibatches with the loop inside the library (single process; under data parallelism the host loop above stays,
 * because the all-reduce sits between its calls).  replaces: the loop body causalbgm/base.py:490-505 for the minibatches
 * perm_dev[0 .. n_use) taken `batch` rows at a time (the last one may be short): bgm_causal_fit_z_sync (lazy = 2), _theta_grad,
 * _theta_apply, _z_step with batch_global = the minibatch's own size, in that order.  loss_dev / loss_z_dev: the accumulators of
 * _theta_grad / _z_step.  With lazy = 1 / 2 on the row-tile chains the latent phase of minibatch k runs on a second stream beside the
 * theta phase of minibatch k + 1 (disjoint rows of the latent table: the minibatches of one call must not share rows -- a permutation);
 * results are those of the sequential order.  The two streams are ordered by counters in device memory that the kernels wait on and
 * advance (csrc/fit_sync.h); a wait that gives up (20 ms bound) voids the call and is reported as BGM_E_HIP by the next
 * bgm_causal_fit_epoch / bgm_causal_fit_end (bgm_bnn_fit_epoch / bgm_bnn_end).  BGM_FIT_NO_FLAGS=1 orders them with HIP events. */
BGM_API int bgm_causal_fit_epoch(bgm_handle *h, const float *x_dev, const float *y_dev, const float *v_dev, float *data_z_dev, float *zm_dev,
                         float *zv_dev, const int32_t *perm_dev, int64_t n_use, int32_t batch, float lr_theta, float lr_z, int32_t lazy,
                         double *loss_dev, double *loss_z_dev, void *stream);

/* The same loop as this rank's share of a DATA-PARALLEL epoch (BASELINE configs[3]; replaces: causalbgm/base.py:488-514 with
 * the rows sharded by observation).  Every rank calls it with ITS rows (x, y, v, data_z and slots, perm_dev over local rows) and the
 * same n_use / batch; a minibatch is `batch` local rows of a global minibatch of batch * world rows (gradients scaled
 * 1 / (b * world)), and between the gradient tiles and the Adam step the fused g|f|h gradient (n_params floats) is summed over the
 * ranks: one ncclAllReduce per step enqueued from C++ on the parameter stream -- no Python between the phases.  All ranks take
 * identical parameter steps; latent rows and their slots stay local.  comm: an ncclComm_t (bgm_comm_create, or the caller's own)
 * handed through as void *.  The latent phase of minibatch k still runs on the second stream beside the theta phase of k + 1; the
 * Adam step is its own launch and the streams are ordered by HIP events.  With a one-rank communicator the results equal
 * bgm_causal_fit_epoch's bit for bit (tests/test_gpu_comm.py). */
BGM_API int bgm_causal_fit_epoch_dp(bgm_handle *h, const float *x_dev, const float *y_dev, const float *v_dev, float *data_z_dev,
                            float *zm_dev, float *zv_dev, const int32_t *perm_dev, int64_t n_use, int32_t batch, float lr_theta,
                            float lr_z, int32_t lazy, double *loss_dev, double *loss_z_dev, void *comm, void *stream);

/* Copy the device parameters of one network back to the host (Keras order) and make them the
 * handle's host copy.  Synchronous. */
BGM_API int bgm_causal_get_weights(bgm_handle *h, int net_id, float *theta_host, int64_t count, void *stream);

/* The gradient half of bgm_causal_fit_z_step: d(batch-mean negative log joint)/d(batch rows of data_z) with the CURRENT networks and
 * the standard-normal latent prior, written to dz_out_dev [batch x q]; no optimizer step.  loss_dev as in bgm_causal_fit_z_step.
 * Used by the host side of IdentifiableCausalBGM (identifiable.py:150-226), which exchanges the prior term for the conditional one
 * and applies its own latent / prior-network updates. */
BGM_API int bgm_causal_fit_z_grad(bgm_handle *h, const float *x_dev, const float *y_dev, const float *v_dev, const float *data_z_dev,
                          const int32_t *idx_dev, int64_t row_lo, int32_t batch, int32_t batch_global, float *dz_out_dev,
                          double *loss_dev, void *stream);

/* Optimizer state of the open fit session, for checkpoints (the reference's tf.train.Checkpoint holds g/f/h_optimizer and
 * posterior_optimizer, causalbgm/base.py:112-122): Adam first / second moments of theta_g | theta_f | theta_h [n_params each] and
 * the step counters steps[0] = theta steps, steps[1] = latent steps.  write = 0 reads them into the host buffers, write = 1
 * installs them (after bgm_causal_fit_begin, which zeroes them).  The latent table's slots zm / zv are the caller's buffers. */
BGM_API int bgm_causal_fit_state(bgm_handle *h, int32_t write, float *m_host, float *v_host, int64_t count, int64_t *steps, void *stream);
/* End the fit session (frees the workspace; the trained parameters stay installed). */
BGM_API int bgm_causal_fit_end(bgm_handle *h, void *stream);

/* ------------------------------------------------------------------------------------------
 * EGM warm start (causalbgm/base.py:305-431): alternating WGAN-GP steps on the latent
 * discriminator dz_net (networks/base.py:338-385) and on g, e, f, h.  One call = one minibatch
 * step = one kernel launch (forward, backward incl. the gradient-penalty double backward, Adam).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t batch_size;                            /* rows per step (egm_init batch_size)            */
  int32_t n_hidden_dz, dz_units[BGM_MAX_LAYERS]; /* params['dz_units']                              */
  float lr;                                      /* params['lr']: Adam(lr, 0.9, 0.99), base.py:86-87 */
  int32_t use_z_rec;                             /* params['use_z_rec'], base.py:367                */
} bgm_egm_config;

/* Start a warm-start session from the networks currently installed with bgm_causal_set_weights (all four:
 * g, e, f, h).  theta_dz_host: discriminator parameters [W0..WL | b0..bL | gamma0.. | beta0..] (W_l row-major
 * [in x out], L = n_hidden_dz hidden layers + the scalar output layer). */
BGM_API int bgm_causal_egm_begin(bgm_handle *h, const bgm_egm_config *cfg, const float *theta_dz_host, int64_t count,
                         void *stream);
/* replaces: train_disc_step, base.py:305-330.  z_dev [B x q] prior sample, idx_dev [B] panel rows,
 * v_dev [N x p] panel, eps = interpolation coefficient (tf.random.uniform([])).  apply = 0 leaves the
 * gradient in the session (bgm_causal_egm_read) without the Adam step.  out_dev: [dz_loss, d_loss] or NULL. */
BGM_API int bgm_causal_egm_disc_step(bgm_handle *h, const float *z_dev, const int32_t *idx_dev, const float *v_dev, float eps,
                             int32_t apply, float *out_dev, void *stream);
/* replaces: train_gen_step, base.py:332-377.  out_dev: [e_loss_adv, l2_loss_v, l2_loss_z, l2_loss_x, l2_loss_y,
 * g_e_loss] or NULL. */
BGM_API int bgm_causal_egm_gen_step(bgm_handle *h, const float *z_dev, const int32_t *idx_dev, const float *v_dev,
                            const float *x_dev, const float *y_dev, int32_t apply, float *out_dev, void *stream);
/* Data-parallel warm start (one process per GPU, every rank holds ITS rows of the panel; the literal collective of north_star,
 * causalbgm/base.py:305-377 under a distribution strategy).  Per step and rank:
 *     bgm_causal_egm_{disc,gen}_step(..., apply = 0, ...)   on the rank's share of the minibatch (session batch_size = B_local)
 *     bgm_causal_egm_grad(which, B_local / B_global, buf)   -> [caller: RCCL all-reduce(SUM) of buf across ranks]
 *     bgm_causal_egm_apply(which, buf)                      the Adam step the apply = 1 call would have taken, from the reduced gradient
 * which = 0: generator side [g | e | f | h] (count = their parameters, Keras order), 1: discriminator.  The means over the minibatch
 * (losses, gradient penalty) become means over the global minibatch; a discriminator with batch statistics (disc_norm = 0) normalises
 * with its rank's rows, as unsynchronised batch normalisation does under any data-parallel scheme. */
BGM_API int bgm_causal_egm_grad(bgm_handle *h, int32_t which, float scale, float *grad_dev, int64_t count, void *stream);
BGM_API int bgm_causal_egm_apply(bgm_handle *h, int32_t which, const float *grad_dev, int64_t count, void *stream);
/* Copy session state to the host: what = 0 generator-side parameters [g | e | f | h] (Keras order), 1 discriminator
 * parameters, 2 / 3 the gradients of the last gen / disc step.  Synchronises the stream. */
BGM_API int bgm_causal_egm_read(bgm_handle *h, int32_t what, float *host, int64_t count, void *stream);
/* Install the session's current g, e, f, h in the handle (as bgm_causal_set_weights would); the session continues. */
BGM_API int bgm_causal_egm_sync(bgm_handle *h, void *stream);
/* bgm_causal_egm_sync + free the session. */
BGM_API int bgm_causal_egm_end(bgm_handle *h, void *stream);

/* ==========================================================================================
 * BGM (bgm/base.py): posterior of Z given partially observed rows, HMC, predictive draws.
 * g_net = BaseVariationalNet (networks/base.py:53-117) evaluated with training=False.
 * ========================================================================================== */
typedef struct {
  int32_t x_dim, z_dim;
  int32_t n_hidden_g, g_units[BGM_MAX_LAYERS];
} bgm_bgm_config;

/* Any g_units in [1, 4096], any depth up to 8, any z_dim are accepted.  g_units = [64]*3 or [64]*5 with z_dim <= 16 run on the
 * register-chained kernels: x_dim in (16,32] and (96,112] with the whole generator resident in LDS, every other width (e.g. config C4,
 * x_dim = 500) with the 2 x 64 x x_dim head weights streamed through an LDS stage shared by the waves of a block (x_dim <= ~2000).
 * Every other shape runs on the general-width engine (csrc/gx_bgm_api.hip; hidden widths up to ~280 fit its LDS tiles). */
BGM_API int bgm_bgm_configure(bgm_handle *h, const bgm_bgm_config *cfg);

/* Generator parameters from HOST memory, flat float32:
 *   BatchNormalization gamma[q], beta[q], moving_mean[q], moving_variance[q]   (networks/base.py:76),
 *   then per trunk Dense layer W [in x out], b [out], then mean_layer W, b, then var_layer W, b. */
BGM_API int bgm_bgm_set_weights(bgm_handle *h, const float *theta_host, int64_t count, void *stream);

/* log p(z | x_obs) + const for n rows; x_dev [n x p] with NaN marking missing cells
 * (the reference passes index lists + obs_mask: bgm/base.py:578-592, 689-700).  grad_dev
 * [n x q] (dlogp/dz) may be NULL.  replaces: BGM.get_log_posterior, bgm/base.py:665-705. */
BGM_API int bgm_bgm_logpost(bgm_handle *h, const float *z_dev, const float *x_dev, int64_t n, float *out_dev,
                    float *grad_dev, void *stream);

/* One segment of HMC transitions for all rows.
 * replaces: tfp.mcmc.HamiltonianMonteCarlo(step_size, num_leapfrog_steps).one_step driven by
 * tfp.mcmc.sample_chain, bgm/base.py:798-821. */
typedef struct {
  const float *x_dev;          /* [n x p], NaN = missing                          */
  int64_t n, row_base;
  float *state_dev;            /* [n x q] in/out                                  */
  float *logp_dev;             /* [n]     in/out (cached target log prob)         */
  float *grad_dev;             /* [n x q] in/out (cached gradient)                */
  int32_t init;                /* 1: state ~ N(0,1) (bgm/base.py:778)             */
  int32_t it_begin, n_iters, burn_in, n_leapfrog;
  const float *step_dev;       /* device scalar: step size                        */
  uint64_t seed;
  double *acc_prob_sum_dev;    /* [>= it_begin+n_iters] += sum_chains exp(min(0, log_accept_ratio)), or NULL */
  uint32_t *acc_count_dev;     /* [>= it_begin+n_iters] += accepted chains, or NULL */
  float *draws_dev;            /* [n_keep x n x q] states after burn_in, or NULL  */
} bgm_hmc_args;
BGM_API int bgm_bgm_hmc_run(bgm_handle *h, const bgm_hmc_args *args, void *stream);
/* Arithmetic of the generator's products in bgm_bgm_logpost / bgm_bgm_hmc_run: 0 = fp32 (default, the reference's arithmetic),
 * 2 = "f16x3": every product (trunk and the two x_dim-wide heads, ~80 % of a gradient evaluation at BASELINE config C4) on
 * v_mfma_f32_16x16x32_f16 with hi / lo fp16 splits of weights, activations and back-propagated gradients, three products per
 * contraction, fp32 accumulation (22 mantissa bits); the whole generator is streamed through LDS as packed fp16 fragments
 * (csrc/bgm_kernels.h "Split precision").  Opt-in; the likelihood, the leapfrog and all sums stay fp32.  Served: trunk
 * [64] x 3 / [64] x 5, z_dim <= 16, any x_dim; BGM_E_UNSUPPORTED for the general-width engine.  fp16 range: weights are clamped to
 * 65504 by the packer, gradients to 6e4 in the kernel.
 * same functions as above: get_log_posterior bgm/base.py:665-705, tfp_mcmc_sampler :798-821. */
BGM_API int bgm_bgm_set_precision(bgm_handle *h, int32_t mode);

/* tfp.mcmc.SimpleStepSizeAdaptation update after iteration `it`:  *step_dev *= (1+rate) if
 * acc_prob_sum_dev[it] / n_chains > target else /= (1+rate).  (bgm/base.py:805-809: target 0.75;
 * TFP default adaptation_rate 0.01.)  With several ranks, all-reduce acc_prob_sum_dev[it] first. */
BGM_API int bgm_bgm_hmc_adapt(bgm_handle *h, float *step_dev, const double *acc_prob_sum_dev, int32_t it,
                      double n_chains, float target, float rate, void *stream);

/* Posterior-predictive draws x ~ N(mu(z_d), sigma^2(z_d)) for draws_dev [n_draws x n x q]:
 * full_dev [n_draws x n x p] (or NULL) and / or cells_dev [(row*k_slots + slot)*n_draws + d] for
 * the cells with slot_dev[row*p + c] >= 0 (or NULL); var_full_dev [n_draws x n x p] receives sigma^2
 * (or NULL); add_noise = 0 returns the mean instead of a draw (use_x_sd=False in generate/evaluate,
 * :470-473,505-508).  replaces: predict_on_posteriors, :511-525; g_net(z, training=False), :468,503. */
BGM_API int bgm_bgm_predict_draws(bgm_handle *h, const float *draws_dev, int64_t n, int64_t row_base, int32_t n_draws,
                          int32_t burn_in, uint64_t seed, const int32_t *slot_dev, int32_t k_slots,
                          float *cells_dev, float *full_dev, float *var_full_dev, int32_t add_noise,
                          void *stream);

/* BGM.fit step functions.  replaces: BGM.update_g_net (bgm/base.py:145-164),
 * BGM.update_latent_variable_sgd (:167-187) and the loop body :399-413, with g_net called with
 * training=True (BatchNormalization on z uses the minibatch statistics and updates its moving
 * averages, networks/base.py:100).  The batch statistics are those of the rows handed to the call (not all-reduced).
 *   theta order = bgm_bgm_set_weights order; the moving statistics receive a zero gradient.
 *   loss_dev (double[4], may be NULL): [0] += sum loss_x, [1] += sum |x-mu|^2 (theta phase),
 *                                      [2] += sum loss_px_z, [3] += sum |x-mu|^2 (z phase). */
BGM_API int bgm_bgm_fit_begin(bgm_handle *h, int64_t n_rows, int32_t max_batch, void *stream);
BGM_API int bgm_bgm_fit_n_params(bgm_handle *h, int64_t *n_params);
/* Data-parallel fit: the batch-mean losses of the following steps are means over `batch_global` rows (the sum of the
 * ranks' local batches), so that the all-reduced SUM of the ranks' gradients is the gradient of the global batch mean.
 * The input BatchNorm of every rank still uses the statistics of its LOCAL batch (a stated deviation from one global
 * batch).  0 (default) = the local batch. */
BGM_API int bgm_bgm_fit_set_global_batch(bgm_handle *h, int32_t batch_global);
BGM_API int bgm_bgm_fit_theta_grad(bgm_handle *h, const float *x_dev, const float *data_z_dev, const int32_t *idx_dev,
                           int32_t batch, float *grad_dev, double *loss_dev, void *stream);
BGM_API int bgm_bgm_fit_theta_apply(bgm_handle *h, const float *grad_dev, float lr_theta, void *stream);
/* Z step: the batch latents are a FRESH variable every minibatch in the reference (:402), i.e. Adam
 * slots start at zero while the step counter keeps running; the updated rows are written back. */
BGM_API int bgm_bgm_fit_z_step(bgm_handle *h, const float *x_dev, float *data_z_dev, const int32_t *idx_dev,
                       int32_t batch, float lr_z, double *loss_dev, void *stream);
/* The minibatch loop of BGM.fit inside the library (single process; under data parallelism the three calls above stay, the all-reduce
 * sits between the first two).  replaces: the loop body bgm/base.py:399-413 for the n_steps minibatches perm_dev[k * batch ..
 * (k + 1) * batch): _theta_grad, _theta_apply, _z_step in that order; loss_dev as in those calls ([4] accumulators). */
BGM_API int bgm_bgm_fit_epoch(bgm_handle *h, const float *x_dev, float *data_z_dev, const int32_t *perm_dev, int64_t n_steps, int32_t batch,
                      float lr_theta, float lr_z, double *loss_dev, void *stream);
BGM_API int bgm_bgm_get_weights(bgm_handle *h, float *theta_host, int64_t count, void *stream);
BGM_API int bgm_bgm_fit_end(bgm_handle *h, void *stream);

/* Measurement aid: effective shader clock (MHz) and fp32-MFMA rate (TFLOP/s) of this device under a
 * back-to-back v_mfma_f32_16x16x4_f32 load on every CU (8 waves/CU, `iters` x 16 MFMAs per wave).
 * Synchronous.  Used by bench.py to state the roofline at the clock the chip actually sustains. */
/* ------------------------------------------------------------------------------------------
 * EGM warm start of BGM (bgm/base.py:190-340): LSGAN discriminators dz_net (latent) and dx_net (data), generator
 * g_net (BaseVariationalNet called with training=True: batch statistics, moving averages updated by every call)
 * and encoder e_net; Adam(lr, beta1 0.5, beta2 0.9), base.py:82-85.  One call = one minibatch step = one launch.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t batch_size;
  int32_t n_hidden_e, e_units[BGM_MAX_LAYERS];   /* params['e_units']                       */
  int32_t n_hidden_dz, dz_units[BGM_MAX_LAYERS]; /* params['dz_units']                      */
  int32_t n_hidden_dx, dx_units[BGM_MAX_LAYERS]; /* params['dx_units']                      */
  float lr;                                      /* params['lr']                            */
  float gamma;                                   /* gradient-penalty weight params['gamma'] */
  float alpha;                                   /* variance regulariser params['alpha']    */
} bgm_bgm_egm_config;

/* Start a session from the generator installed with bgm_bgm_set_weights.  theta_e: encoder [W0,b0,...] (x_dim ->
 * e_units -> z_dim); theta_dz / theta_dx: discriminators [W0..WL | b0..bL | gamma.. | beta..]. */
BGM_API int bgm_bgm_egm_begin(bgm_handle *h, const bgm_bgm_egm_config *cfg, const float *theta_e_host, int64_t count_e,
                      const float *theta_dz_host, int64_t count_dz, const float *theta_dx_host, int64_t count_dx,
                      void *stream);
/* replaces: train_disc_step, bgm/base.py:190-244.  z_dev [B x z_dim] prior sample, x_dev [B x x_dim] data rows,
 * noise_dev [B x x_dim] standard normals of the reparameterisation, eps_* the interpolation coefficients of the
 * gradient penalties (used when gamma != 0).  out_dev: [dz_loss, dx_loss, d_loss] or NULL. */
BGM_API int bgm_bgm_egm_disc_step(bgm_handle *h, const float *z_dev, const float *x_dev, const float *noise_dev, float eps_z,
                          float eps_x, int32_t apply, float *out_dev, void *stream);
/* replaces: train_gen_step, bgm/base.py:246-289.  out_dev: [g_loss_adv, e_loss_adv, l2_loss_z, l2_loss_x, reg_loss,
 * g_e_loss] or NULL. */
BGM_API int bgm_bgm_egm_gen_step(bgm_handle *h, const float *z_dev, const float *x_dev, const float *noise1_dev,
                         const float *noise2_dev, int32_t apply, float *out_dev, void *stream);
/* Session state <-> host: what = 0 generator side [g (bgm_bgm_set_weights order) | e], 1 discriminators [dz | dx],
 * 2 / 3 gradients of the last gen / disc step (read only). */
BGM_API int bgm_bgm_egm_read(bgm_handle *h, int32_t what, float *host, int64_t count, void *stream);
BGM_API int bgm_bgm_egm_write(bgm_handle *h, int32_t what, const float *host, int64_t count, void *stream);
/* z = e_net(x) for n rows with the session's encoder (Z initialisation bgm/base.py:384, evaluate :466). */
BGM_API int bgm_bgm_egm_encode(bgm_handle *h, const float *x_dev, int64_t n, float *z_dev, void *stream);
/* Install the session's generator in the handle (as bgm_bgm_set_weights would); _end also frees the session. */
BGM_API int bgm_bgm_egm_sync(bgm_handle *h, void *stream);
BGM_API int bgm_bgm_egm_end(bgm_handle *h, void *stream);

/* ==========================================================================================
 * CausalBGM with Bayesian networks, params['use_bnn'] = True (the default of every causal YAML and of the CLI).
 * g, e, f, h = BayesianFullyConnectedNet (networks/bnn.py:4-38): BatchNormalization of the input on the
 * statistics of the batch at hand + tfp.layers.DenseFlipout stack; every call draws fresh weight perturbations.
 * Semantics and the counter-based noise streams: oracle/bnn.py.  A session owns the parameters
 *   theta = [g | e | f | h], per net: gamma[in], beta[in], then per layer loc[in x out], rho[in x out], bias[out]
 * (sigma = finfo(f32).eps + softplus(rho)), their Adam slots and a gradient buffer of the same layout.
 * Noise of a call: key = seed + (batch_id << 32), stream = 32-bit call id (see each entry point).
 * ========================================================================================== */
#define BGM_BNN_G 0
#define BGM_BNN_E 1
#define BGM_BNN_F 2
#define BGM_BNN_H 3
typedef struct {
  int32_t v_dim, z_dims[4], binary_treatment;    /* params['v_dim'], ['z_dims'], ['binary_treatment']            */
  int32_t n_hidden[4];                           /* hidden layers of g, e, f, h (BGM_BNN_* order)                */
  int32_t units[4][BGM_MAX_LAYERS];              /* params['g_units'], ['e_units'], ['f_units'], ['h_units']      */
  float kl_weight;                               /* params['kl_weight'], base.py:171-173                          */
  int32_t max_batch;                             /* largest minibatch of the step kernels (<= 256)                */
  int32_t norm_mode;                             /* input BatchNormalization: 0 = statistics of the batch at hand (the
                                                    reference as written: inner layer called inside call(training=True));
                                                    1 = fixed mean 0 / variance 1 (inference mode on never-updated moving
                                                    averages: the alternative reading, keeps a constant treatment column) */
  float sigma_v, sigma_x, sigma_y;               /* params['sigma_v' | 'sigma_x' | 'sigma_y'] (base.py:161,195,224,257,268,283,698): > 0 = the
                                                    fixed standard deviation of that likelihood (the net's variance head is then
                                                    neither read nor trained by the data terms); 0 = the variance head                 */
} bgm_bnn_config;

/* Open a session.  theta_host: `count` floats in the layout above (count from bgm_bnn_layout). */
BGM_API int bgm_bnn_begin(bgm_handle *h, const bgm_bnn_config *cfg, const float *theta_host, int64_t count, void *stream);
/* Parameter count of a configuration and the offsets of g, e, f, h in theta (offsets[4] = total).  No session needed. */
BGM_API int bgm_bnn_layout(const bgm_bnn_config *cfg, int64_t offsets[5]);
/* what = 0 parameters, 1 gradient of the last step, 2 Adam first moments, 3 Adam second moments.  Synchronises. */
BGM_API int bgm_bnn_read(bgm_handle *h, int32_t what, float *host, int64_t count, void *stream);
BGM_API int bgm_bnn_write(bgm_handle *h, int32_t what, const float *host, int64_t count, void *stream);
/* replaces: update_g_net, update_h_net, update_f_net with use_bnn, causalbgm/base.py:156-243 (one launch: the three
 * updates are independent given the batch).  data_z_dev [N x q]; idx_dev [batch] rows of the panel x_dev [N], y_dev [N],
 * v_dev [N x p].  Losses are batch means over batch_global rows (= batch on one GPU) + kl_weight * sum(KL).
 * Noise stream `stream_id`.  apply = 1: Adam(lr_theta, 0.9, 0.99) on g, h, f in the same launch; apply = 0: the gradient
 * stays in the session (bgm_bnn_read what = 1 / bgm_bnn_grad_dev) for an all-reduce, then bgm_bnn_theta_apply.
 * out_dev: [loss_v, loss_mse_v, loss_x, loss_mse_x|bce, loss_y, loss_mse_y] or NULL. */
BGM_API int bgm_bnn_theta_step(bgm_handle *h, const float *data_z_dev, const int32_t *idx_dev, const float *x_dev,
                       const float *y_dev, const float *v_dev, int32_t batch, int32_t batch_global, float lr_theta,
                       uint64_t seed, uint32_t stream_id, int32_t apply, float *out_dev, void *stream);
BGM_API int bgm_bnn_grad_dev(bgm_handle *h, float **grad_dev, int64_t *count);
/* Copy the session's gradient to buf_dev (to_session = 0) or back (1): [n_params] floats, device to device. */
BGM_API int bgm_bnn_grad_exchange(bgm_handle *h, float *buf_dev, int32_t to_session, void *stream);
BGM_API int bgm_bnn_theta_apply(bgm_handle *h, float lr_theta, void *stream);
/* replaces: update_latent_variable_sgd with use_bnn, base.py:246-302 (every net called twice with independent noise:
 * streams stream_id and stream_id + 1) + the Adam step on the latent table (zm_dev, zv_dev: its slots, [n_rows x q];
 * lazy = 0: Keras dense-decay semantics, every row of the table moves; lazy = 1: batch rows only; lazy = 2: lazy = 0 with the
 * untouched rows' steps deferred, see bgm_bnn_z_sync).  out_dev: [loss_postrior_z] or NULL. */
BGM_API int bgm_bnn_z_step(bgm_handle *h, const float *x_dev, const float *y_dev, const float *v_dev, float *data_z_dev,
                   float *zm_dev, float *zv_dev, const int32_t *idx_dev, int64_t n_rows, int32_t batch,
                   int32_t batch_global, float lr_z, int32_t lazy, uint64_t seed, uint32_t stream_id, float *out_dev,
                   float *dz_out_dev, void *stream);
/* Replay mode of the latent optimizer with Bayesian nets; semantics and call order as bgm_causal_fit_z_sync (idx_dev rows before the
 * minibatch's bgm_bnn_theta_step calls; idx_dev = NULL flushes the whole [n_rows x q] table). */
BGM_API int bgm_bnn_z_sync(bgm_handle *h, float *data_z_dev, float *zm_dev, float *zv_dev, const int32_t *idx_dev, int64_t n_rows,
                   int32_t batch, float lr_z, void *stream);
/* A whole list of minibatches with the loop inside the library (single process): for the minibatches perm_dev[0 .. n_use) taken
 * `batch` rows at a time (a one-row tail is skipped) bgm_bnn_z_sync (lazy = 2), bgm_bnn_theta_step(apply = 1, noise stream
 * stream_id0 + 3 k) and bgm_bnn_z_step (streams stream_id0 + 3 k + 1, + 2) in that order; *n_done = minibatches run (the caller advances
 * its stream counter by 3 * n_done).  replaces: the loop body causalbgm/base.py:490-505 with use_bnn.  With lazy = 1 / 2 on the
 * row-tile chains the latent phase of minibatch k runs on a second stream beside the forward / backward chains of minibatch k + 1
 * (the minibatches of one call must not share rows); results are those of the sequential order. */
BGM_API int bgm_bnn_fit_epoch(bgm_handle *h, const float *x_dev, const float *y_dev, const float *v_dev, float *data_z_dev, float *zm_dev,
                      float *zv_dev, const int32_t *perm_dev, int64_t n_rows, int64_t n_use, int32_t batch, float lr_theta, float lr_z,
                      int32_t lazy, uint64_t seed, uint32_t stream_id0, float *out_t_dev, float *out_z_dev, int32_t *n_done, void *stream);
/* This rank's share of a data-parallel epoch with Bayesian nets: per minibatch bgm_bnn_z_sync (lazy = 2), bgm_bnn_theta_step(apply = 0,
 * batch_global = b * world), ONE ncclAllReduce of the session's fused gradient (in place, on `stream`), bgm_bnn_theta_apply,
 * bgm_bnn_z_step -- the host loop of models/causalbgm_bnn.py issued from C++.  comm as in bgm_causal_fit_epoch_dp. */
BGM_API int bgm_bnn_fit_epoch_dp(bgm_handle *h, const float *x_dev, const float *y_dev, const float *v_dev, float *data_z_dev, float *zm_dev,
                         float *zv_dev, const int32_t *perm_dev, int64_t n_rows, int64_t n_use, int32_t batch, float lr_theta,
                         float lr_z, int32_t lazy, uint64_t seed, uint32_t stream_id0, float *out_t_dev, float *out_z_dev,
                         int32_t *n_done, void *comm, void *stream);
BGM_API int bgm_bnn_end(bgm_handle *h, void *stream);

/* ---- EGM warm start with Bayesian nets (train_disc_step :305-330, train_gen_step :332-377 with use_bnn): a sub-session of
 * bgm_bnn_begin that trains the session's g, e, f, h (own Adam slots = g_pre_optimizer) against the deterministic latent
 * discriminator dz_net (parameter layout and bgm_egm_config as in bgm_causal_egm_begin).  Noise: the encoder call of the
 * disc step uses stream stream_id; the nine network calls of the gen step use stream_id + 0..8 in the order
 * g(z), g(z) [variance head], e(v), e(v_), g(z_), f, f [variance head], h, h [variance head]. */
BGM_API int bgm_bnn_egm_begin(bgm_handle *h, const bgm_egm_config *cfg, const float *theta_dz_host, int64_t count, void *stream);
BGM_API int bgm_bnn_egm_disc_step(bgm_handle *h, const float *z_dev, const int32_t *idx_dev, const float *v_dev, float eps,
                          uint64_t seed, uint32_t stream_id, int32_t apply, float *out_dev, void *stream);
BGM_API int bgm_bnn_egm_gen_step(bgm_handle *h, const float *z_dev, const int32_t *idx_dev, const float *v_dev, const float *x_dev,
                         const float *y_dev, uint64_t seed, uint32_t stream_id, int32_t apply, float *out_dev, void *stream);
/* Data-parallel form, as bgm_causal_egm_grad / _apply: which = 0 the session's Bayesian nets (count = bgm_bnn_begin's parameter count,
 * gradient incl. the KL terms, identical on every rank), 1 the discriminator.  All ranks pass the same (seed, stream_id): one weight
 * perturbation per step for the global minibatch, as Flipout has it. */
BGM_API int bgm_bnn_egm_set_share(bgm_handle *h, int32_t row0);   /* this rank's rows are rows row0 .. row0 + batch_size - 1 of the global
                                                                    * minibatch: the Flipout sign vector of its row b is the one of global row row0 + b */
BGM_API int bgm_bnn_egm_grad(bgm_handle *h, int32_t which, float scale, float *grad_dev, int64_t count, void *stream);
BGM_API int bgm_bnn_egm_apply(bgm_handle *h, int32_t which, const float *grad_dev, int64_t count, void *stream);
/* what = 1 discriminator parameters, 3 its gradient of the last disc step (the nets' side: bgm_bnn_read). */
BGM_API int bgm_bnn_egm_read(bgm_handle *h, int32_t what, float *host, int64_t count, void *stream);
BGM_API int bgm_bnn_egm_end(bgm_handle *h, void *stream);

/* ---- large-batch side of the session: posterior sampling, causal effects, evaluation.  A "block" is the batch of rows
 * one reference call sees (bs rows of predict, base.py:640-645; the whole panel for evaluate): its input statistics
 * normalise the rows and ONE weight perturbation per layer is shared by them.  Noise key of block b = seed + (b << 32). */

/* replaces: get_log_posterior with use_bnn, base.py:765-817.  Rows [0, n) in blocks of block_rows (first block id
 * block0); one call of g, h, f per block with noise stream stream_id.  out_dev [n]. */
BGM_API int bgm_bnn_logpost(bgm_handle *h, const float *x_dev, const float *y_dev, const float *v_dev, const float *z_dev,
                    int64_t n, int32_t block_rows, int32_t block0, uint64_t seed, uint32_t stream_id, float *out_dev,
                    void *stream);

typedef struct {
  const float *x_dev, *y_dev, *v_dev;  /* [n], [n], [n x p]                                                       */
  int64_t n, row_base;                 /* rows of this call; global index of row 0 (keys the per-row RNG streams)  */
  int32_t block_rows, block0;          /* bs of predict; id of the first block                                     */
  float *state_dev;                    /* [n x q] chain states (in / out)                                          */
  int32_t init;                        /* 1: draw the initial states N(0, I) (base.py:842) before it_begin         */
  int32_t it_begin, n_iters, burn_in;  /* iterations [it_begin, it_begin + n_iters); kept when it >= burn_in       */
  float q_sd;                          /* proposal std-dev (fixed)                                                 */
  uint64_t seed;
  uint32_t *acc_count_dev;             /* optional: accepted proposals (accumulated)                               */
  float *draws_dev;                    /* optional [n_keep x n x q]: retained states                               */
  int32_t n_keep;
  int32_t effect;                      /* 0 none, 1 ADRF (continuous), 2 ITE (binary)                              */
  int32_t sample_y;
  const float *x_values_dev;           /* effect 1: [n_doses]                                                      */
  int32_t n_doses;
  double *adrf_sum_dev;                /* effect 1: [n_doses x n_keep] sums over the rows of this call (accumulated) */
  float *ite_dev;                      /* effect 2: [n x n_keep]                                                    */
  const float *q_sd_blocks_dev;        /* optional [n_blocks]: proposal std-dev per block (overrides q_sd): every block
                                          of the reference is its own sampler run with its own adaptive scale          */
  uint32_t *acc_blocks_dev;            /* optional [n_iters x n_blocks]: accepted proposals per iteration and block
                                          (accumulated; the caller's sliding acceptance window, base.py:873-884)       */
  int64_t block_row0;                  /* position of this call's first row INSIDE block block0 (0: the call starts a block).  > 0: the call
                                          is a rank's share of ONE block (it must end inside it): the Flipout sign words are keyed by
                                          (block, position in the block), so the result does not depend on how a block's rows are split
                                          over ranks (IdentifiableCausalBGM.predict: the whole panel is one block, identifiable.py:557-614).
                                          Default-shape sampling kernels only (BGM_E_UNSUPPORTED elsewhere)              */
} bgm_bnn_mh_args;
/* replaces: metropolis_hastings_sampler (fixed q_sd) + infer_from_latent_posterior with use_bnn, base.py:820-904,
 * 671-763.  All blocks advance in lock step, three launches per iteration (perturbations, proposal + statistics,
 * g/h/f forward of both states + accept).  Noise streams: 2 it (proposal), 2 it + 1 (current state);
 * effects of kept draw d at dose k: stream 0x40000000 + d * n_doses + k (ITE: k = 0 for x = 1, 1 for x = 0). */
BGM_API int bgm_bnn_mh_run(bgm_handle *h, const bgm_bnn_mh_args *args, void *stream);
/* replaces: infer_from_latent_posterior with use_bnn, base.py:671-763, on a given draw tensor draws_dev [n_keep x n x q]
 * (the stand-alone form; bgm_bnn_mh_run computes the same quantities for the draws it keeps).  Draw d uses the outcome-noise
 * iteration it0 + d and the Flipout streams 0x40000000 + d * n_doses + k.  effect = 1: adrf_sum_dev [n_doses x n_keep] (fp64,
 * sums over the n rows, accumulated); effect = 2: ite_dev [n x n_keep]. */
BGM_API int bgm_bnn_effects(bgm_handle *h, const float *draws_dev, int64_t n, int32_t block_rows, int32_t block0, int64_t row_base,
                    int32_t n_keep, int32_t it0, uint64_t seed, int32_t effect, int32_t sample_y, const float *x_values_dev,
                    int32_t n_doses, double *adrf_sum_dev, float *ite_dev, void *stream);
/* replaces: evaluate with use_bnn, base.py:534-570 (the whole panel of n rows is ONE batch), and the Z initialisation
 * data_z = e_net(data_v) of fit, base.py:479.  encode = 1: z_dev [n x q] is WRITTEN with e(v) first (noise stream
 * stream_id); otherwise it is read.  sums_dev (optional, fp64 [3]): sums over rows of (v - v^)^2 (all p columns),
 * (x - x^)^2, (y - y^)^2 from one call of g, h, f (stream stream_id).  dose_sums_dev (continuous, fp64 [n_doses]): sums
 * over rows of mu_y at the doses x_values_dev (stream stream_id + 1 + k).  ite_dev (binary, [n]): mu_y(x = 1) - mu_y(x = 0)
 * (streams stream_id + 1, + 2).  Any of the three outputs may be NULL. */
BGM_API int bgm_bnn_evaluate(bgm_handle *h, const float *x_dev, const float *y_dev, const float *v_dev, float *z_dev, int32_t encode,
                     int64_t n, const float *x_values_dev, int32_t n_doses, uint64_t seed, uint32_t stream_id,
                     double *sums_dev, double *dose_sums_dev, float *ite_dev, void *stream);

/* ==========================================================================================
 * BGM with the Bayesian generator (params['use_bnn'] = True; bgm/base.py:67-69).
 * g_net = BayesianVariationalNet (networks/bnn.py:40-99): BatchNormalization that honours `training`, DenseFlipout trunk +
 * LeakyReLU(0.2), two sibling DenseFlipout heads (mean, softplus variance + 1e-6); kernel AND bias prior N(0, 0.1^2).
 * Flat parameter layout: gamma[q], beta[q], moving_mean[q], moving_variance[q], then loc [in x out], rho [in x out],
 * bias [out] per layer in the order trunk..., mean head, var head.  Noise: oracle/bgm_bnn.py (key = seed, one stream id per
 * generator call).  Every call perturbs the kernels, also with training=False.
 * ========================================================================================== */
typedef struct {
  int32_t x_dim, z_dim;
  int32_t n_hidden_g, g_units[BGM_MAX_LAYERS];
  float kl_weight;               /* params['kl_weight'] */
  int32_t max_batch;             /* largest minibatch of the step functions (2..64) */
  int32_t hmc_frozen_noise;      /* 0: every gradient evaluation of HMC draws a fresh perturbation (the reference as written);
                                    1: the whole HMC run reuses generator call 0 (one weight draw, deterministic target) */
} bgm_bvn_config;
BGM_API int bgm_bvn_layout(const bgm_bvn_config *cfg, int64_t *n_params);
/* Open a session with the parameters theta_host (layout above).  Adam slots start at zero. */
BGM_API int bgm_bvn_begin(bgm_handle *h, const bgm_bvn_config *cfg, const float *theta_host, int64_t count, void *stream);
/* what = 0 parameters, 1 gradient of the last theta step, 2 / 3 Adam slots */
BGM_API int bgm_bvn_read(bgm_handle *h, int32_t what, float *host, int64_t count, void *stream);
BGM_API int bgm_bvn_write(bgm_handle *h, int32_t what, const float *host, int64_t count, void *stream);
/* replaces: BGM.update_g_net with use_bnn, bgm/base.py:145-164 (loss_x + kl_weight * sum(g_net.losses); training-mode
 * BatchNormalization incl. its moving averages).  apply = 1: Adam(lr, 0.9, 0.99) inside the call; apply = 0: the gradient
 * (of the mean over batch_global rows) stays in the session for bgm_bvn_grad_exchange / bgm_bvn_theta_apply.
 * out_dev [2] = loss_x, loss_mse (may be NULL). */
BGM_API int bgm_bvn_theta_step(bgm_handle *h, const float *x_dev, float *data_z_dev, const int32_t *idx_dev, int32_t batch,
                       int32_t batch_global, float lr, uint64_t seed, uint32_t stream_id, int32_t apply, float *out_dev,
                       void *stream);
BGM_API int bgm_bvn_grad_exchange(bgm_handle *h, float *buf_dev, int32_t to_session, void *stream);
BGM_API int bgm_bvn_theta_apply(bgm_handle *h, float lr, void *stream);
/* replaces: BGM.update_latent_variable_sgd with use_bnn, bgm/base.py:167-187, and the fresh-slot Adam step on the batch rows
 * of data_z (:402).  out_dev [1] = loss_postrior_z. */
BGM_API int bgm_bvn_z_step(bgm_handle *h, const float *x_dev, float *data_z_dev, const int32_t *idx_dev, int32_t batch,
                   int32_t batch_global, float lr_z, uint64_t seed, uint32_t stream_id, float *out_dev, void *stream);
/* replaces: BGM.get_log_posterior with use_bnn, bgm/base.py:665-705: ONE generator call (stream_id) over the n rows (global
 * rows row_base + i key the Flipout signs); x_dev NaN = missing; grad_dev [n x q] may be NULL. */
BGM_API int bgm_bvn_logpost(bgm_handle *h, const float *z_dev, const float *x_dev, int64_t n, int64_t row_base, uint64_t seed,
                    uint32_t stream_id, float *out_dev, float *grad_dev, void *stream);
/* replaces: tfp.mcmc.HamiltonianMonteCarlo.one_step on the stochastic target (bgm/base.py:798-821): gradient evaluation
 * `leap` of transition `it` is generator call 1 + it * n_leapfrog + leap (call 0 = bootstrap when init = 1); the cached
 * log-prob / gradient of the current state are kept, as TFP does.  Same argument struct as bgm_bgm_hmc_run. */
BGM_API int bgm_bvn_hmc_run(bgm_handle *h, const bgm_hmc_args *args, void *stream);
/* Arithmetic of bgm_bvn_hmc_run launched afterwards (no reference counterpart: the reference computes in fp32): 0 fp32 (default) |
 * 2 f16x3 -- the generator's products (posterior means and perturbation) in split fp16 with fp32 accumulation (bgmfx_kernels.h);
 * opt-in.  f16x3 serves frozen-noise HMC (cfg.hmc_frozen_noise) of generators with 3 or 5 hidden layers of 64 units and
 * z_dim <= 16; anything else: BGM_E_UNSUPPORTED and the mode is unchanged. */
BGM_API int bgm_bvn_set_precision(bgm_handle *h, int32_t mode);
/* replaces: g_net(z, training=False) + reparameterize in predict_on_posteriors / generate / evaluate (:511-525, :478-509,
 * :444-476): ONE generator call (stream_id) over the flattened [n_draws x n] rows; the Flipout signs of draw d, row r are
 * keyed by d * sign_stride + sign_off + r (predict: sign_stride = bs, sign_off = position of the first row inside its
 * bs-block, so the result does not depend on how a block is split over ranks).  Outputs as bgm_bgm_predict_draws. */
BGM_API int bgm_bvn_decode(bgm_handle *h, const float *draws_dev, int64_t n, int64_t row_base, int32_t n_draws, int32_t burn_in,
                   uint64_t seed, uint32_t stream_id, uint32_t sign_stride, uint32_t sign_off, const int32_t *slot_dev,
                   int32_t k_slots, float *cells_dev, float *full_dev, float *var_full_dev, int32_t add_noise, void *stream);
/* EGM warm start with the Bayesian generator (bgm/base.py:190-340; e_net, dz_net, dx_net deterministic): as bgm_bgm_egm_* but
 * on top of the bgm_bvn session, whose generator it copies at _begin and returns at _sync / _end.  (seed, stream_id) key the
 * Flipout noise: the disc step makes one generator call (stream_id), the gen step two (stream_id, stream_id + 1). */
BGM_API int bgm_bvn_egm_begin(bgm_handle *h, const bgm_bgm_egm_config *cfg, const float *theta_e_host, int64_t count_e,
                      const float *theta_dz_host, int64_t count_dz, const float *theta_dx_host, int64_t count_dx,
                      void *stream);
BGM_API int bgm_bvn_egm_disc_step(bgm_handle *h, const float *z_dev, const float *x_dev, const float *noise_dev, float eps_z,
                          float eps_x, uint64_t seed, uint32_t stream_id, int32_t apply, float *out_dev, void *stream);
BGM_API int bgm_bvn_egm_gen_step(bgm_handle *h, const float *z_dev, const float *x_dev, const float *noise1_dev,
                         const float *noise2_dev, uint64_t seed, uint32_t stream_id, int32_t apply, float *out_dev,
                         void *stream);
BGM_API int bgm_bvn_egm_read(bgm_handle *h, int32_t what, float *host, int64_t count, void *stream);
BGM_API int bgm_bvn_egm_write(bgm_handle *h, int32_t what, const float *host, int64_t count, void *stream);
BGM_API int bgm_bvn_egm_encode(bgm_handle *h, const float *x_dev, int64_t n, float *z_dev, void *stream);
BGM_API int bgm_bvn_egm_sync(bgm_handle *h, void *stream);
BGM_API int bgm_bvn_egm_end(bgm_handle *h, void *stream);
BGM_API int bgm_bvn_end(bgm_handle *h, void *stream);

/* ---- RCCL communicator for the data-parallel minibatch loops (csrc/comm_api.hip).  The reference has no collective; this is
 * the exchange north_star names: "the outer loop over observations shards data-parallel across the 8 GPUs of one node with an RCCL
 * all-reduce of generator / discriminator gradients over xGMI".  RCCL is resolved at run time from the copy the process already maps
 * (PyTorch-ROCm's librccl.so.1), else the system one (BGM_RCCL_LIB overrides); the calls answer BGM_E_UNSUPPORTED with the reason when
 * none is found.  Rendezvous: rank 0 calls bgm_comm_unique_id, the BGM_COMM_ID_BYTES bytes travel to the other ranks by any means the
 * host side has (torch.distributed broadcast in bayesgm_amd/parallel.py), every rank calls bgm_comm_create(its device, id, world, rank). */
#define BGM_COMM_ID_BYTES 128
BGM_API int bgm_comm_unique_id(void *id_out);
BGM_API int bgm_comm_create(int32_t device, const void *id, int32_t world, int32_t rank, void **comm_out);
BGM_API int bgm_comm_destroy(void *comm);
/* ranks / this process's rank of a communicator, and which librccl serves it (library: caller's buffer, may be NULL) */
BGM_API int bgm_comm_info(void *comm, int32_t *world, int32_t *rank, char *library, int32_t library_cap);
/* in-place sum of `count` floats over the ranks, enqueued on `stream` (what the *_fit_epoch_dp loops issue per minibatch; exported
 * so that tests and bench.py can time the collective alone) */
BGM_API int bgm_comm_all_reduce_f32(void *comm, float *buf_dev, int64_t count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BGM_HIP_H */
