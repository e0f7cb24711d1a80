// bnn_sample_api.hip -- C ABI of the Bayesian-network CausalBGM path: posterior sampling, effects, evaluation.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "bgm_host.h"
#include "bnf_host.h"
#include "bnn_sample_kernels.h"
#include "bnn_state.h"

static BnnState *bst(bgm_handle *h) { return static_cast<BnnState *>(h->bnn_state); }

template <class K>
static int bns_set_lds(K kernel, int bytes) {
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return BGM_OK;
}

struct BnsPlan {
  BnsNet net[4];
  long long frag_total = 0;      // packed loc / sigma floats (all four nets)
  long long set_ghf = 0;         // floats of one perturbation set [g | h | f]
  long long set_f = 0;           // ... of the outcome net alone
  long long set_all = 0;         // ... [e | g | h | f] (evaluation)
  int lds_bytes = 0;
  int eff_lds_bytes = 0;       // effects kernel: + the second perturbation buffer of its pipelined dose loop
};

// fragment plan of the session's nets; false when a shape is outside the kernels' limits
static bool bns_plan(const BnnState *s, BnsPlan &pl) {
  long long fb = 0;
  int maxfrag = 0;
  for (int k = 0; k < 4; ++k) {
    BnsNet &n = pl.net[k];
    std::memset(&n, 0, sizeof(n));
    bns_from(s->net[k], n);
    if (n.n_layers < 2 || n.swords > BNS_SW || n.K[0] > BNS_MAXK) return false;
    for (int l = 0; l < n.n_layers; ++l) {
      if (l < n.n_layers - 1 && n.K[l + 1] > 16 * BNS_MAXT) return false;

      maxfrag = std::max(maxfrag, n.T[l] * n.MT[l] * 256);
    }
    { int bsum = 0; for (int l = 0; l < n.n_layers; ++l) bsum += 16 * n.MT[l]; if (bsum > BNS_MAXB) return false; }
    n.fbase = (int)fb;
    fb += n.foff[n.n_layers];
  }
  pl.frag_total = fb;
  pl.net[BNN_G].dbase = 0;
  pl.net[BNN_H].dbase = pl.net[BNN_G].foff[pl.net[BNN_G].n_layers];
  pl.net[BNN_F].dbase = pl.net[BNN_H].dbase + pl.net[BNN_H].foff[pl.net[BNN_H].n_layers];
  pl.set_ghf = pl.net[BNN_F].dbase + pl.net[BNN_F].foff[pl.net[BNN_F].n_layers];
  pl.net[BNN_E].dbase = (int)pl.set_ghf;     // evaluation sets: [g | h | f | e]
  pl.set_all = pl.set_ghf + pl.net[BNN_E].foff[pl.net[BNN_E].n_layers];
  pl.set_f = pl.net[BNN_F].foff[pl.net[BNN_F].n_layers];
  (void)maxfrag;
  pl.lds_bytes = (int)sizeof(float) * (2 * BNS_MAXK + BNS_MAXB + BNS_WAVES * BNS_R * 16 * BNS_SW + 2 * BNS_CHUNK * 256);
  pl.eff_lds_bytes = pl.lds_bytes + (int)sizeof(float) * BNS_CHUNK * 256;
  return true;
}

// effects kernel: the pipelined dose loop when the outcome net's shape allows it (bns_eff_fast_ok), else the generic routine
static auto bns_eff_kernel(const BnsEffArgs &ea) -> void (*)(BnsEffArgs) {
  return bns_eff_default_ok(ea.f) ? bns_effects_kernel<2> : bns_eff_fast_ok(ea.f) ? bns_effects_kernel<1> : bns_effects_kernel<0>;
}
static int bns_set_lds_eff(const BnsPlan &pl) {
  int rc = bns_set_lds(bns_effects_kernel<2>, pl.eff_lds_bytes);
  if (!rc) rc = bns_set_lds(bns_effects_kernel<1>, pl.eff_lds_bytes);
  return rc ? rc : bns_set_lds(bns_effects_kernel<0>, pl.eff_lds_bytes);
}

// Device scratch of the sampling side: [lf | sf | zprop | dw sets | stats (doubles) | xstats (doubles)], grown on demand.
struct BnsBuf { float *lf, *sf, *zprop, *dw, *pair; double *stats, *xstats, *vstats; };
static int bns_buffers(bgm_handle *h, BnnState *s, const BnsPlan &pl, long long n, int n_blocks, long long dw_floats, BnsBuf &b,
                       hipStream_t stream) {
  const long long fr = (pl.frag_total + 63) & ~63LL;
  const long long zp = (n * s->q + 63) & ~63LL;
  const long long dwf = (dw_floats + 63) & ~63LL;
  const long long st_d = 2LL * n_blocks * 256 + 2LL * n_blocks + 2 * BNS_MAXK + 64;      // doubles
  const size_t need = (size_t)(2 * fr + zp + dwf + 2 * st_d + 128);
  if (need > s->samp_cap) {
    BGM_HIP_CHECK(hipStreamSynchronize(stream));
    if (s->samp_dev) BGM_HIP_CHECK(hipFree(s->samp_dev));
    s->samp_dev = nullptr; s->samp_cap = 0; s->packed_valid = false;
    BGM_HIP_CHECK(hipMalloc((void **)&s->samp_dev, sizeof(float) * need));
    s->samp_cap = need;
  }
  b.lf = s->samp_dev; b.sf = b.lf + fr; b.zprop = b.sf + fr; b.dw = b.zprop + zp;
  b.stats = (double *)(b.dw + dwf);
  b.xstats = b.stats + 2LL * n_blocks * 256;
  b.vstats = b.xstats + 2LL * n_blocks;
  b.pair = (float *)(b.stats + st_d);
  if (!s->packed_valid) {
    BGM_HIP_CHECK(hipMemsetAsync(b.lf, 0, sizeof(float) * 2 * fr, stream));
    BnsPackArgs pa{};
    for (int k = 0; k < 4; ++k) pa.net[k] = pl.net[k];
    pa.theta = s->theta_dev; pa.lf = b.lf; pa.sf = b.sf; pa.n_nets = 4;
    hipLaunchKernelGGL(bns_pack_kernel, dim3(32, 4), dim3(256), 0, stream, pa);
    BGM_HIP_CHECK(hipGetLastError());
    s->packed_valid = true;
  }
  // padding positions of the perturbation sets must be zero: the noise kernel only writes real elements
  BGM_HIP_CHECK(hipMemsetAsync(b.dw, 0, sizeof(float) * dwf, stream));
  BGM_HIP_CHECK(hipMemsetAsync(b.stats, 0, sizeof(double) * st_d, stream));
  return BGM_OK;
}

static thread_local bool bns_wide = false;
static int bns_session(bgm_handle *h, const char *who, BnnState *&s, BnsPlan &pl) {
  if (!h || !h->bnn_state) { bgm_set_error(std::string(who) + ": no session (bgm_bnn_begin)"); return BGM_E_STATE; }
  s = bst(h);
  bns_wide = !bns_plan(s, pl);      // outside the LDS-fragment kernels (hidden widths <= 64, inputs <= 208): the any-width path, bnw_api.hip
  return BGM_OK;
}

static void bns_fill_mh(const BnnState *s, const BnsPlan &pl, BnsMhArgs &a) {
  for (int k = 0; k < 4; ++k) a.net[k] = pl.net[k];
  a.theta = s->theta_dev;
  a.q = s->q; a.p = s->p;
  a.z0 = s->cfg.z_dims[0]; a.z1 = s->cfg.z_dims[1]; a.z2 = s->cfg.z_dims[2];
  a.binary = s->cfg.binary_treatment;
  a.set_floats = pl.set_ghf;
  a.sig2[0] = s->cfg.sigma_v > 0.0f ? s->cfg.sigma_v * s->cfg.sigma_v : 0.0f;
  a.sig2[1] = s->cfg.sigma_x > 0.0f ? s->cfg.sigma_x * s->cfg.sigma_x : 0.0f;
  a.sig2[2] = s->cfg.sigma_y > 0.0f ? s->cfg.sigma_y * s->cfg.sigma_y : 0.0f;
}

static void bns_noise(const BnsPlan &pl, const BnsBuf &b, const int *ids, int n_nets, int n_sets_blocks, int n_calls, long long set_floats,
                      uint64_t seed, uint32_t stream0, uint32_t stride, int block0, hipStream_t stream) {
  BnsNoiseArgs na{};
  for (int i = 0; i < n_nets; ++i) na.net[i] = pl.net[ids[i]];
  na.n_nets = n_nets; na.n_calls = n_calls; na.sf = b.sf; na.dw = b.dw; na.set_floats = set_floats;
  na.k0 = (uint32_t)(seed & 0xFFFFFFFFull); na.k1 = (uint32_t)(seed >> 32); na.stream0 = stream0; na.stream_stride = stride;
  na.block0 = block0;
  hipLaunchKernelGGL(bns_noise_kernel, dim3(8, n_sets_blocks * n_calls), dim3(256), 0, stream, na);
}

extern "C" int bgm_bnn_logpost(bgm_handle *h, const float *x, const float *y, const float *v, const float *z, int64_t n,
                               int32_t block_rows, int32_t block0, uint64_t seed, uint32_t stream_id, float *out, void *stream_) {
  BnnState *s; BnsPlan pl;
  int rc = bns_session(h, "bgm_bnn_logpost", s, pl);
  if (rc) return rc;
  if (!x || !y || !v || !z || !out || n < 1 || block_rows < 2) { bgm_set_error("bgm_bnn_logpost: bad argument"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  rc = bnf_logpost(h, s, x, y, v, z, n, block_rows, block0, seed, stream_id, out, stream);     // inference-mode normalisation, default shapes
  if (rc <= 0) return rc;
  // any hidden width: the any-width path (it reads the conditional prior's row tables as well)
  if (bns_wide) return bnw_logpost(h, s, x, y, v, z, n, block_rows, block0, seed, stream_id, out, stream);
  const int n_blocks = (int)((n + block_rows - 1) / block_rows);
  BnsBuf b;
  rc = bns_buffers(h, s, pl, n, n_blocks, (long long)n_blocks * pl.set_ghf, b, stream);
  if (rc) return rc;
  const int ids[3] = {BNN_G, BNN_H, BNN_F};
  bns_noise(pl, b, ids, 3, n_blocks, 1, pl.set_ghf, seed, stream_id, 0, block0, stream);
  BnsPropArgs pa{};
  pa.z = z; pa.zprop = b.zprop; pa.n = n; pa.row_base = 0; pa.q = s->q; pa.bs = block_rows;
  pa.wg_per_block = (block_rows + 255) / 256; pa.it = 0; pa.init = 0; pa.q_sd = 0.0f;
  pa.k0 = (uint32_t)seed; pa.k1 = (uint32_t)(seed >> 32);
  pa.stats = b.stats; pa.n_blocks = n_blocks; pa.par = 0; pa.x = x; pa.xstats = b.xstats;
  hipLaunchKernelGGL(bns_propose_kernel, dim3(n_blocks * pa.wg_per_block), dim3(256), 0, stream, pa);
  BnsMhArgs a{};
  bns_fill_mh(s, pl, a);
  a.lf = b.lf; a.dw = b.dw; a.stats = b.stats; a.xstats = b.xstats;
  a.x = x; a.y = y; a.v = v; a.z = const_cast<float *>(z); a.zprop = b.zprop; a.n = n; a.row_base = 0;
  a.bs = block_rows; a.wg_per_block = (block_rows + BNS_ROWS - 1) / BNS_ROWS; a.block0 = block0; a.mode = 0; a.it = 0;
  a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.stream0 = stream_id; a.out = out;
  if (s->bp_on) {      // conditional prior under batch statistics: one noisy call of the prior net for this evaluation (bprior_api.hip)
    if ((rc = bprior_rows(h, s, n, block_rows, block0, seed, stream_id, 1, stream))) return rc;
    a.prior = s->bp_rows; a.prior_stride = 0;
  }
  rc = bns_set_lds(bns_mh_kernel, pl.lds_bytes);
  if (rc) return rc;
  a.n_items = n_blocks * a.wg_per_block;
  hipLaunchKernelGGL(bns_mh_kernel, dim3((a.n_items + 7) & ~7), dim3(BNS_THREADS), pl.lds_bytes, stream, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bnn_mh_run(bgm_handle *h, const bgm_bnn_mh_args *g, void *stream_) {
  BnnState *s; BnsPlan pl;
  int rc = bns_session(h, "bgm_bnn_mh_run", s, pl);
  if (rc) return rc;
  if (!g || !g->x_dev || !g->y_dev || !g->v_dev || !g->state_dev || g->n < 1 || g->block_rows < 2 || g->n_iters < 0) {
    bgm_set_error("bgm_bnn_mh_run: bad argument"); return BGM_E_INVALID;
  }
  if (g->effect == 1 && (!g->x_values_dev || g->n_doses < 1 || !g->adrf_sum_dev)) { bgm_set_error("bgm_bnn_mh_run: ADRF needs x_values_dev and adrf_sum_dev"); return BGM_E_INVALID; }
  if (g->effect == 2 && !g->ite_dev) { bgm_set_error("bgm_bnn_mh_run: ITE needs ite_dev"); return BGM_E_INVALID; }
  if (g->effect < 0 || g->effect > 2) { bgm_set_error("bgm_bnn_mh_run: bad effect"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  rc = bnf_mh_run(h, s, g, stream);
  if (rc <= 0) return rc;
  if (g->block_row0 != 0) { bgm_set_error("bgm_bnn_mh_run: a share of one block (block_row0 > 0) is served by the default-shape sampling kernels with inference-mode normalisation only"); return BGM_E_UNSUPPORTED; }
  if (bns_wide) return bnw_mh_run(h, s, g, stream);
  const long long n = g->n;
  const int bs = g->block_rows, n_blocks = (int)((n + bs - 1) / bs), q = s->q;
  BnsBuf b;
  const int n_doses = g->effect == 1 ? g->n_doses : (g->effect == 2 ? 2 : 0);
  rc = bns_buffers(h, s, pl, n, n_blocks, 2LL * n_blocks * pl.set_ghf + (long long)n_blocks * n_doses * pl.set_f, b, stream);
  if (rc) return rc;
  rc = bns_set_lds(bns_mh_kernel, pl.lds_bytes);
  if (rc) return rc;
  rc = bns_set_lds_eff(pl);
  if (rc) return rc;
  float *dw_eff = b.dw + 2LL * n_blocks * pl.set_ghf;
  BnsEffArgs ea{};
  if (g->effect) {
    if (g->effect == 2) {
      static const float pair_host[2] = {1.0f, 0.0f};
      BGM_HIP_CHECK(hipMemcpyAsync(b.pair, pair_host, sizeof(pair_host), hipMemcpyHostToDevice, stream));
    }
    ea.f = pl.net[BNN_F]; ea.f.dbase = 0; ea.sig2_y = s->cfg.sigma_y > 0.0f ? s->cfg.sigma_y * s->cfg.sigma_y : 0.0f;
    ea.theta = s->theta_dev; ea.lf = b.lf; ea.dw = dw_eff; ea.set_floats = pl.set_f;
    ea.z = g->state_dev; ea.n = n; ea.row_base = g->row_base; ea.q = q; ea.z0 = s->cfg.z_dims[0]; ea.z1 = s->cfg.z_dims[1];
    ea.bs = bs; ea.wg_per_block = (bs + BNS_ROWS - 1) / BNS_ROWS; ea.block0 = g->block0; ea.n_doses = n_doses;
    ea.xvals = g->effect == 1 ? g->x_values_dev : b.pair;
    ea.k0 = (uint32_t)g->seed; ea.k1 = (uint32_t)(g->seed >> 32); ea.sample_y = g->sample_y;
  }
  // effects of the draw kept by iteration `it_kept`, whose state statistics sit in slot 1 of parity `par`
  auto effects = [&](int it_kept, int par) {
    const int d = it_kept - g->burn_in;
    BnsNoiseArgs na{};
    na.net[0] = ea.f; na.n_nets = 1; na.n_calls = n_doses; na.sf = b.sf; na.dw = dw_eff; na.set_floats = pl.set_f;
    na.k0 = ea.k0; na.k1 = ea.k1; na.stream0 = 0x40000000u + (uint32_t)d * (uint32_t)n_doses; na.stream_stride = 1u; na.block0 = g->block0;
    hipLaunchKernelGGL(bns_noise_kernel, dim3(2, n_blocks * n_doses), dim3(256), 0, stream, na);
    ea.stats = b.stats + (long long)par * n_blocks * 256;
    ea.stream0 = na.stream0; ea.it_noise = (uint32_t)it_kept;
    ea.sum_out = g->effect == 1 ? g->adrf_sum_dev + d : nullptr; ea.sum_stride = g->n_keep;
    ea.ite_out = g->effect == 2 ? g->ite_dev + d : nullptr; ea.ite_stride = g->n_keep;
    ea.n_items = n_blocks * ea.wg_per_block;
    hipLaunchKernelGGL(bns_eff_kernel(ea), dim3((ea.n_items + 7) & ~7), dim3(BNS_THREADS), pl.eff_lds_bytes, stream, ea);
  };
  auto kept = [&](int it) { return g->effect && it >= g->burn_in && it - g->burn_in < g->n_keep; };
  const int ids[3] = {BNN_G, BNN_H, BNN_F};
  BnsPropArgs pa{};
  pa.z = g->state_dev; pa.zprop = b.zprop; pa.n = n; pa.row_base = g->row_base; pa.q = q; pa.bs = bs;
  pa.wg_per_block = (bs + 255) / 256; pa.q_sd = g->q_sd; pa.q_sd_blocks = g->q_sd_blocks_dev;
  pa.k0 = (uint32_t)g->seed; pa.k1 = (uint32_t)(g->seed >> 32);
  pa.stats = b.stats; pa.n_blocks = n_blocks; pa.x = g->x_dev; pa.z_init = g->state_dev;
  BnsMhArgs a{};
  bns_fill_mh(s, pl, a);
  a.lf = b.lf; a.dw = b.dw; a.xstats = b.xstats;
  a.x = g->x_dev; a.y = g->y_dev; a.v = g->v_dev; a.z = g->state_dev; a.zprop = b.zprop; a.n = n; a.row_base = g->row_base;
  a.bs = bs; a.wg_per_block = (bs + BNS_ROWS - 1) / BNS_ROWS; a.block0 = g->block0; a.mode = 1;
  a.k0 = pa.k0; a.k1 = pa.k1; a.acc_count = g->acc_count_dev;
#ifdef BNS_PROF
  static unsigned long long *prof_dev = nullptr;
  if (!prof_dev) hipMalloc((void **)&prof_dev, 128);
  hipMemsetAsync(prof_dev, 0, 128, stream);
  a.prof = prof_dev;
  ea.prof = prof_dev + 8;
#endif
  for (int i = 0; i < g->n_iters; ++i) {
    const int it = g->it_begin + i;
    bns_noise(pl, b, ids, 3, n_blocks, 2, pl.set_ghf, g->seed, 2u * (uint32_t)it, 1u, g->block0, stream);
    pa.it = it; pa.par = i & 1; pa.init = (i == 0 && g->init) ? 1 : 0; pa.xstats = (i == 0) ? b.xstats : nullptr;
    hipLaunchKernelGGL(bns_propose_kernel, dim3(n_blocks * pa.wg_per_block), dim3(256), 0, stream, pa);
    if (i > 0 && kept(it - 1)) effects(it - 1, i & 1);
    if (s->bp_on) {      // the two evaluations' own calls of the prior net (streams 2 it, 2 it + 1)
      if ((rc = bprior_rows(h, s, n, bs, g->block0, g->seed, 2u * (uint32_t)it, 2, stream))) return rc;
      a.prior = s->bp_rows; a.prior_stride = n * (long long)(q + 2);
    }
    a.it = it; a.stats = b.stats + (long long)(i & 1) * n_blocks * 256;
    a.acc_blocks = g->acc_blocks_dev ? g->acc_blocks_dev + (long long)i * n_blocks : nullptr;
    a.n_items = n_blocks * a.wg_per_block;
  hipLaunchKernelGGL(bns_mh_kernel, dim3((a.n_items + 7) & ~7), dim3(BNS_THREADS), pl.lds_bytes, stream, a);
    if (g->draws_dev && it >= g->burn_in && it - g->burn_in < g->n_keep)
      BGM_HIP_CHECK(hipMemcpyAsync(g->draws_dev + (long long)(it - g->burn_in) * n * q, g->state_dev, sizeof(float) * n * q,
                                   hipMemcpyDeviceToDevice, stream));
  }
#ifdef BNS_PROF
  {
    unsigned long long hp[6];
    hipStreamSynchronize(stream);
    hipMemcpy(hp, prof_dev, sizeof(hp), hipMemcpyDeviceToHost);
    const double wgs = (double)n_blocks * a.wg_per_block * std::max(1, g->n_iters);
    int occ = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bns_mh_kernel, BNS_THREADS, pl.lds_bytes);
    fprintf(stderr, "[BNS_PROF] occupancy %d workgroups / CU (lds %d B)\n", occ, pl.lds_bytes);
    unsigned long long he[6];
    hipMemcpy(he, prof_dev + 8, sizeof(he), hipMemcpyDeviceToHost);
    if (g->effect) fprintf(stderr, "[BNS_PROF] effects, cycles per workgroup-launch: prologue %.0f staging %.0f first %.0f middle %.0f last %.0f outside %.0f\n",
            he[0] / wgs, he[1] / wgs, he[2] / wgs, he[3] / wgs, he[4] / wgs, he[5] / wgs);
    fprintf(stderr, "[BNS_PROF] cycles per workgroup-iteration (wave 0): prologue %.0f staging %.0f first %.0f middle %.0f last %.0f outside %.0f\n",
            hp[0] / wgs, hp[1] / wgs, hp[2] / wgs, hp[3] / wgs, hp[4] / wgs, hp[5] / wgs);
  }
#endif
  if (g->n_iters > 0 && kept(g->it_begin + g->n_iters - 1)) {
    // statistics of the final state: one more statistics pass (its proposal is not used)
    const int i = g->n_iters;
    pa.it = g->it_begin + i; pa.par = i & 1; pa.init = 0; pa.xstats = nullptr;
    hipLaunchKernelGGL(bns_propose_kernel, dim3(n_blocks * pa.wg_per_block), dim3(256), 0, stream, pa);
    effects(g->it_begin + i - 1, i & 1);
  }
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bnn_evaluate(bgm_handle *h, const float *x, const float *y, const float *v, float *z, int32_t encode, int64_t n,
                                const float *x_values, int32_t n_doses, uint64_t seed, uint32_t stream_id, double *sums,
                                double *dose_sums, float *ite, void *stream_) {
  BnnState *s; BnsPlan pl;
  int rc = bns_session(h, "bgm_bnn_evaluate", s, pl);
  if (rc) return rc;
  if (!v || !z || n < 2 || n > 0x7FFFFFFFLL) { bgm_set_error("bgm_bnn_evaluate: bad argument"); return BGM_E_INVALID; }
  const bool recon = sums != nullptr;
  if (recon && (!x || !y)) { bgm_set_error("bgm_bnn_evaluate: x_dev, y_dev required"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  const int binary = s->cfg.binary_treatment;
  const int nd = dose_sums ? n_doses : (ite ? 2 : 0);
  if (dose_sums && (!x_values || n_doses < 1)) { bgm_set_error("bgm_bnn_evaluate: dose grid missing"); return BGM_E_INVALID; }
  if (bns_wide) return bnw_evaluate(h, s, x, y, v, z, encode, n, x_values, n_doses, seed, stream_id, sums, dose_sums, ite, stream);
  BnsBuf b;
  rc = bns_buffers(h, s, pl, n, 1, pl.set_all + (long long)nd * pl.set_f, b, stream);
  if (rc) return rc;
  rc = bns_set_lds(bns_eval_kernel, pl.lds_bytes);
  if (rc) return rc;
  rc = bns_set_lds_eff(pl);
  if (rc) return rc;
  const int ids[4] = {BNN_G, BNN_H, BNN_F, BNN_E};
  bns_noise(pl, b, ids, 4, 1, 1, pl.set_all, seed, stream_id, 0, 0, stream);
  BnsEvalArgs ev{};
  for (int k = 0; k < 4; ++k) ev.net[k] = pl.net[k];
  ev.theta = s->theta_dev; ev.lf = b.lf; ev.dw = b.dw; ev.stats = b.stats; ev.xstats = b.xstats; ev.vstats = b.vstats;
  ev.x = x; ev.y = y; ev.v = v; ev.z = z; ev.n = n; ev.q = s->q; ev.p = s->p;
  ev.z0 = s->cfg.z_dims[0]; ev.z1 = s->cfg.z_dims[1]; ev.z2 = s->cfg.z_dims[2]; ev.binary = binary;
  ev.k0 = (uint32_t)seed; ev.k1 = (uint32_t)(seed >> 32); ev.stream = stream_id; ev.sums = sums;
  const int wgs = (int)((n + BNS_ROWS - 1) / BNS_ROWS);
  if (encode) {
    hipLaunchKernelGGL(bns_colstats_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 256), s->p), dim3(256), 0, stream, v, (long long)n,
                       s->p, b.vstats);
    ev.mode = 0;
    hipLaunchKernelGGL(bns_eval_kernel, dim3(wgs), dim3(BNS_THREADS), pl.lds_bytes, stream, ev);
  }
  if (recon || nd) {
    // statistics of z (slot 1) and of x
    BnsPropArgs pa{};
    pa.z = z; pa.zprop = b.zprop; pa.n = n; pa.row_base = 0; pa.q = s->q; pa.bs = (int)n; pa.wg_per_block = (int)((n + 255) / 256);
    pa.q_sd = 0.0f; pa.k0 = ev.k0; pa.k1 = ev.k1; pa.stats = b.stats; pa.n_blocks = 1; pa.par = 0; pa.x = x; pa.xstats = x ? b.xstats : nullptr;
    hipLaunchKernelGGL(bns_propose_kernel, dim3(pa.wg_per_block), dim3(256), 0, stream, pa);
  }
  if (recon) {
    BGM_HIP_CHECK(hipMemsetAsync(sums, 0, 3 * sizeof(double), stream));
    ev.mode = 1;
    hipLaunchKernelGGL(bns_eval_kernel, dim3(wgs), dim3(BNS_THREADS), pl.lds_bytes, stream, ev);
  }
  if (nd) {
    float *dw_eff = b.dw + pl.set_all;
    BnsEffArgs ea{};
    ea.f = pl.net[BNN_F]; ea.f.dbase = 0; ea.sig2_y = s->cfg.sigma_y > 0.0f ? s->cfg.sigma_y * s->cfg.sigma_y : 0.0f;
    if (!dose_sums) {
      static const float pair_host[2] = {1.0f, 0.0f};
      BGM_HIP_CHECK(hipMemcpyAsync(b.pair, pair_host, sizeof(pair_host), hipMemcpyHostToDevice, stream));
    } else {
      BGM_HIP_CHECK(hipMemsetAsync(dose_sums, 0, sizeof(double) * nd, stream));
    }
    BnsNoiseArgs na{};
    na.net[0] = ea.f; na.n_nets = 1; na.n_calls = nd; na.sf = b.sf; na.dw = dw_eff; na.set_floats = pl.set_f;
    na.k0 = ev.k0; na.k1 = ev.k1; na.stream0 = stream_id + 1u; na.stream_stride = 1u; na.block0 = 0;
    hipLaunchKernelGGL(bns_noise_kernel, dim3(2, nd), dim3(256), 0, stream, na);
    ea.theta = s->theta_dev; ea.lf = b.lf; ea.dw = dw_eff; ea.set_floats = pl.set_f; ea.stats = b.stats;
    ea.z = z; ea.n = n; ea.row_base = 0; ea.q = s->q; ea.z0 = ev.z0; ea.z1 = ev.z1; ea.bs = (int)n; ea.wg_per_block = wgs; ea.block0 = 0;
    ea.n_doses = nd; ea.xvals = dose_sums ? x_values : b.pair; ea.k0 = ev.k0; ea.k1 = ev.k1; ea.stream0 = stream_id + 1u;
    ea.sample_y = 0; ea.sum_out = dose_sums; ea.sum_stride = 1; ea.ite_out = dose_sums ? nullptr : ite; ea.ite_stride = 1;
    ea.n_items = wgs;
    hipLaunchKernelGGL(bns_eff_kernel(ea), dim3((wgs + 7) & ~7), dim3(BNS_THREADS), pl.eff_lds_bytes, stream, ea);
  }
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

extern "C" int bgm_bnn_effects(bgm_handle *h, const float *draws, int64_t n, int32_t block_rows, int32_t block0, int64_t row_base,
                               int32_t n_keep, int32_t it0, uint64_t seed, int32_t effect, int32_t sample_y, const float *x_values,
                               int32_t n_doses, double *adrf_sum, float *ite, void *stream_) {
  BnnState *s; BnsPlan pl;
  int rc = bns_session(h, "bgm_bnn_effects", s, pl);
  if (rc) return rc;
  if (!draws || n < 1 || block_rows < 2 || n_keep < 1 || (effect != 1 && effect != 2)) { bgm_set_error("bgm_bnn_effects: bad argument"); return BGM_E_INVALID; }
  if (effect == 1 && (!x_values || n_doses < 1 || !adrf_sum)) { bgm_set_error("bgm_bnn_effects: ADRF needs x_values_dev and adrf_sum_dev"); return BGM_E_INVALID; }
  if (effect == 2 && !ite) { bgm_set_error("bgm_bnn_effects: ITE needs ite_dev"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  rc = bnf_effects(h, s, draws, n, block_rows, block0, row_base, n_keep, it0, seed, effect, sample_y, x_values, n_doses, adrf_sum, ite, stream);
  if (rc <= 0) return rc;
  if (bns_wide) return bnw_effects(h, s, draws, n, block_rows, block0, row_base, n_keep, it0, seed, effect, sample_y, x_values, n_doses, adrf_sum, ite, stream);
  const int bs = block_rows, n_blocks = (int)((n + bs - 1) / bs), q = s->q;
  const int nd = effect == 1 ? n_doses : 2;
  BnsBuf b;
  rc = bns_buffers(h, s, pl, n, n_blocks, (long long)n_blocks * nd * pl.set_f, b, stream);
  if (rc) return rc;
  rc = bns_set_lds_eff(pl);
  if (rc) return rc;
  if (effect == 2) {
    static const float pair_host[2] = {1.0f, 0.0f};
    BGM_HIP_CHECK(hipMemcpyAsync(b.pair, pair_host, sizeof(pair_host), hipMemcpyHostToDevice, stream));
  }
  BnsEffArgs ea{};
  ea.f = pl.net[BNN_F]; ea.f.dbase = 0; ea.sig2_y = s->cfg.sigma_y > 0.0f ? s->cfg.sigma_y * s->cfg.sigma_y : 0.0f;
  ea.theta = s->theta_dev; ea.lf = b.lf; ea.dw = b.dw; ea.set_floats = pl.set_f;
  ea.n = n; ea.row_base = row_base; ea.q = q; ea.z0 = s->cfg.z_dims[0]; ea.z1 = s->cfg.z_dims[1];
  ea.bs = bs; ea.wg_per_block = (bs + BNS_ROWS - 1) / BNS_ROWS; ea.block0 = block0; ea.n_doses = nd;
  ea.xvals = effect == 1 ? x_values : b.pair;
  ea.k0 = (uint32_t)seed; ea.k1 = (uint32_t)(seed >> 32); ea.sample_y = sample_y;
  BnsPropArgs pa{};
  pa.zprop = b.zprop; pa.n = n; pa.row_base = row_base; pa.q = q; pa.bs = bs; pa.wg_per_block = (bs + 255) / 256; pa.q_sd = 0.0f;
  pa.k0 = ea.k0; pa.k1 = ea.k1; pa.stats = b.stats; pa.n_blocks = n_blocks;
  for (int d = 0; d < n_keep; ++d) {
    const float *z = draws + (long long)d * n * q;
    pa.z = z; pa.it = d; pa.par = d & 1;                      // statistics of this draw (slot 1 of parity d & 1)
    hipLaunchKernelGGL(bns_propose_kernel, dim3(n_blocks * pa.wg_per_block), dim3(256), 0, stream, pa);
    BnsNoiseArgs na{};
    na.net[0] = ea.f; na.n_nets = 1; na.n_calls = nd; na.sf = b.sf; na.dw = b.dw; na.set_floats = pl.set_f;
    na.k0 = ea.k0; na.k1 = ea.k1; na.stream0 = 0x40000000u + (uint32_t)d * (uint32_t)nd; na.stream_stride = 1u; na.block0 = block0;
    hipLaunchKernelGGL(bns_noise_kernel, dim3(2, n_blocks * nd), dim3(256), 0, stream, na);
    ea.z = z; ea.stats = b.stats + (long long)(d & 1) * n_blocks * 256;
    ea.stream0 = na.stream0; ea.it_noise = (uint32_t)(it0 + d);
    ea.sum_out = effect == 1 ? adrf_sum + d : nullptr; ea.sum_stride = n_keep;
    ea.ite_out = effect == 2 ? ite + d : nullptr; ea.ite_stride = n_keep;
    ea.n_items = n_blocks * ea.wg_per_block;
    hipLaunchKernelGGL(bns_eff_kernel(ea), dim3((ea.n_items + 7) & ~7), dim3(BNS_THREADS), pl.eff_lds_bytes, stream, ea);
  }
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}
