// bnw_api.hip -- host side of the any-width sampling / evaluation path of CausalBGM with Bayesian networks (bnw_kernels.h).  Entered
// from the bgm_bnn_* entry points (bnn_sample_api.hip) when no LDS-fragment family holds the session's shape.
#include <algorithm>
#include <cstring>
#include <string>

#include "bgm_host.h"
#include "bnn_state.h"
#include "bnw_kernels.h"

struct BnwState {
  float *dw = nullptr, *ws = nullptr, *pair = nullptr, *loct = nullptr;
  BnwNets *nets = nullptr;      // [2]: the nets of the call's sets, the outcome net's own
  size_t dw_cap = 0, ws_cap = 0, loct_cap = 0;
  float *st = nullptr;          // batch statistics (params['bnn_norm'] = "batch"), doubles: [2 parities][n_blocks][256] | x [n_blocks][2] | v [2][p]
  size_t st_cap = 0;
};

void bnw_free(void *p) {
  BnwState *b = static_cast<BnwState *>(p);
  if (!b) return;
  if (b->dw) hipFree(b->dw);
  if (b->ws) hipFree(b->ws);
  if (b->pair) hipFree(b->pair);
  if (b->loct) hipFree(b->loct);
  if (b->nets) hipFree(b->nets);
  if (b->st) hipFree(b->st);
  delete b;
}

namespace {
int bnw_need(BnnState *s, const char *who) {
  if (s->cfg.norm_mode != 1 && s->q > 64) {
    bgm_set_error(std::string(who) + ": batch statistics (params['bnn_norm'] = 'batch') hold at most 64 latent columns");
    return BGM_E_UNSUPPORTED;
  }
  for (int k = 0; k < 4; ++k)
    if (s->net[k].n_layers < 1) { bgm_set_error(std::string(who) + ": bad network"); return BGM_E_UNSUPPORTED; }
  return BGM_OK;
}
// sets of this launch hold the nets listed in ids (order = layout)
void bnw_nets(const BnnState *s, const int *ids, int n_ids, BnwNets &m) {
  std::memset(&m, 0, sizeof(m));
  for (int k = 0; k < 4; ++k) { m.net[k] = s->net[k]; m.net[k].bn_fixed = 1; m.noff[k] = -1; }      // ("batch": every call is given its block's statistics, BnwStats)
  long long off = 0;
  for (int i = 0; i < n_ids; ++i) { m.noff[ids[i]] = (int)off; off += (s->net[ids[i]].eoff[s->net[ids[i]].n_layers] + 3) & ~3; }
  m.set_floats = off;
  m.theta = s->theta_dev;
  m.q = s->q; m.p = s->p; m.z0 = s->cfg.z_dims[0]; m.z1 = s->cfg.z_dims[1]; m.z2 = s->cfg.z_dims[2]; m.binary = s->cfg.binary_treatment;
  m.sig2[0] = s->cfg.sigma_v > 0.0f ? s->cfg.sigma_v * s->cfg.sigma_v : 0.0f;
  m.sig2[1] = s->cfg.sigma_x > 0.0f ? s->cfg.sigma_x * s->cfg.sigma_x : 0.0f;
  m.sig2[2] = s->cfg.sigma_y > 0.0f ? s->cfg.sigma_y * s->cfg.sigma_y : 0.0f;
}
int bnw_grow(float **p, size_t *cap, size_t floats, hipStream_t stream) {
  if (floats <= *cap) return BGM_OK;
  BGM_HIP_CHECK(hipStreamSynchronize(stream));
  if (*p) BGM_HIP_CHECK(hipFree(*p));
  *p = nullptr; *cap = 0;
  BGM_HIP_CHECK(hipMalloc((void **)p, sizeof(float) * floats));
  *cap = floats;
  return BGM_OK;
}
#define BNW_LDS_BYTES (sizeof(float) * 2 * BNW_STAGE_FLOATS)      // the double-buffered stage of bnw_gemm2
struct BnwPlan { BnwState *b; int grid; long long ws_stride; };
int bnw_plan(bgm_handle *h, BnnState *s, const BnwNets &m, long long n_sets_total, int n_items, BnwPlan &pl, hipStream_t stream) {
  if (!s->bnw) s->bnw = new BnwState();
  BnwState *b = static_cast<BnwState *>(s->bnw);
  pl.b = b;
  pl.grid = std::max(1, std::min(n_items, 4 * h->n_cus));
  pl.ws_stride = (long long)((bnw_ws_floats(m) + 63) & ~(size_t)63);
  int rc = bnw_grow(&b->ws, &b->ws_cap, (size_t)pl.ws_stride * (size_t)(4 * h->n_cus), stream);
  if (rc) return rc;
  rc = bnw_grow(&b->dw, &b->dw_cap, (size_t)n_sets_total * (size_t)m.set_floats + 64, stream);
  if (rc) return rc;
  rc = bnw_grow(&b->loct, &b->loct_cap, 2 * (size_t)m.set_floats + 128, stream);      // the call's sets and the outcome net's own sets
  if (rc) return rc;
  if (!b->nets) {
    BGM_HIP_CHECK(hipMalloc((void **)&b->nets, 2 * sizeof(BnwNets)));
    BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bnw_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BNW_LDS_BYTES));
    BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(bnw_effects_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BNW_LDS_BYTES));
  }
  if (!b->pair) {
    BGM_HIP_CHECK(hipMalloc((void **)&b->pair, 2 * sizeof(float)));
    static const float pair_host[2] = {1.0f, 0.0f};
    BGM_HIP_CHECK(hipMemcpy(b->pair, pair_host, sizeof(pair_host), hipMemcpyHostToDevice));
  }
  return BGM_OK;
}
// transposed posterior means of the nets of m's sets at loct + at (the launches of one call read them; theta may change between calls)
int bnw_pack(BnwState *b, BnwNets &m, size_t at, hipStream_t stream) {
  float *dst = b->loct + at;
  long long mx = 0;
  for (int k = 0; k < 4; ++k) if (m.noff[k] >= 0) mx = std::max<long long>(mx, m.net[k].eoff[m.net[k].n_layers]);
  hipLaunchKernelGGL(bnw_pack_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>(256, (mx + 255) / 256))), dim3(256), 0, stream, m, dst);
  BGM_HIP_CHECK(hipGetLastError());
  m.locT = dst;
  m.dev = b->nets + (at ? 1 : 0);      // the kernels' copy of the nets (device memory: bnw_kernels.h BnwRowsArgs::mp)
  static_assert(sizeof(BnwNets) % 4 == 0, "BnwNets is copied in words");
  hipLaunchKernelGGL(bnw_store_nets_kernel, dim3(1), dim3(64), 0, stream, m, b->nets + (at ? 1 : 0));
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}
// batch statistics: zeroed buffers of this call; stats(par) = the parity a launch accumulates into
struct BnwBatch {
  bool on = false;
  double *stats = nullptr, *xstats = nullptr, *vstats = nullptr;
  int n_blocks = 0;
  const double *parity(int par) const { return stats + (long long)par * n_blocks * 256; }
};
int bnw_batch(BnnState *s, BnwState *b, int n_blocks, BnwBatch &bt, hipStream_t stream) {
  bt.on = s->cfg.norm_mode != 1;
  bt.n_blocks = n_blocks;
  if (!bt.on) return BGM_OK;
  const size_t doubles = (size_t)2 * n_blocks * 256 + (size_t)2 * n_blocks + (size_t)2 * s->p + 16;
  int rc = bnw_grow(&b->st, &b->st_cap, 2 * doubles, stream);
  if (rc) return rc;
  bt.stats = reinterpret_cast<double *>(b->st);
  bt.xstats = bt.stats + (size_t)2 * n_blocks * 256;
  bt.vstats = bt.xstats + (size_t)2 * n_blocks;
  BGM_HIP_CHECK(hipMemsetAsync(b->st, 0, sizeof(double) * doubles, stream));
  return BGM_OK;
}
// column sums of the states z (slot 1) and of their proposals of iteration `it` (slot 0) into parity `par`; with_x: the treatment column too
void bnw_stats(const BnwBatch &bt, const float *z, long long n, long long row_base, int q, int bs, int it, int init, float q_sd, const float *q_sd_blocks,
               uint64_t seed, int par, const float *x, bool with_x, hipStream_t stream) {
  BnwStatArgs sa{};
  sa.z = z; sa.n = n; sa.row_base = row_base; sa.q = q; sa.bs = bs; sa.wg_per_block = (bs + 255) / 256; sa.it = it; sa.init = init;
  sa.q_sd = q_sd; sa.q_sd_blocks = q_sd_blocks; sa.k0 = (uint32_t)seed; sa.k1 = (uint32_t)(seed >> 32);
  sa.stats = bt.stats; sa.n_blocks = bt.n_blocks; sa.par = par; sa.x = x; sa.xstats = with_x ? bt.xstats : nullptr;
  hipLaunchKernelGGL(bnw_stats_kernel, dim3((unsigned)(bt.n_blocks * sa.wg_per_block)), dim3(256), 0, stream, sa);
}
void bnw_noise(const BnwNets &m, float *dw, int n_blocks, int n_calls, int block0, uint64_t seed, uint32_t stream0, uint32_t stride, hipStream_t stream) {
  BnwNoiseArgs na{};
  na.m = m; na.dw = dw; na.n_calls = n_calls; na.block0 = block0;
  na.k0 = (uint32_t)seed; na.k1 = (uint32_t)(seed >> 32); na.stream0 = stream0; na.stream_stride = stride;
  long long mx = 0;
  for (int k = 0; k < 4; ++k) if (m.noff[k] >= 0) mx = std::max<long long>(mx, m.net[k].eoff[m.net[k].n_layers]);
  const int gx = (int)std::max<long long>(1, std::min<long long>(64, (mx / 4 + 255) / 256));
  hipLaunchKernelGGL(bnw_noise_kernel, dim3(gx, n_blocks * n_calls), dim3(256), 0, stream, na);
}
}  // namespace

int bnw_logpost(bgm_handle *h, BnnState *s, const float *x, const float *y, const float *v, const float *z, int64_t n, int32_t bs, int32_t block0,
                uint64_t seed, uint32_t stream_id, float *out, hipStream_t stream) {
  int rc = bnw_need(s, "bgm_bnn_logpost");
  if (rc) return rc;
  const int ids[3] = {BNN_G, BNN_H, BNN_F};
  BnwRowsArgs a{};
  bnw_nets(s, ids, 3, a.m);
  const int n_blocks = (int)((n + bs - 1) / bs), tpb = (bs + BNW_RT - 1) / BNW_RT;
  BnwPlan pl;
  rc = bnw_plan(h, s, a.m, n_blocks, n_blocks * tpb, pl, stream);
  if (rc) return rc;
  if ((rc = bnw_pack(pl.b, a.m, 0, stream))) return rc;
  a.mp = a.m.dev;
  bnw_noise(a.m, pl.b->dw, n_blocks, 1, block0, seed, stream_id, 0u, stream);
  BnwBatch bt;
  rc = bnw_batch(s, pl.b, n_blocks, bt, stream);
  if (rc) return rc;
  if (bt.on) {                          // the call's batch = a block of rows: its statistics (slot 1 = the states themselves)
    bnw_stats(bt, z, n, 0, s->q, bs, 0, 0, 0.0f, nullptr, seed, 0, x, true, stream);
    a.stats = bt.parity(0); a.xstats = bt.xstats;
  }
  if (s->bp_on) {                       // conditional prior: one noisy call of the prior net for this evaluation (bprior_api.hip)
    if ((rc = bprior_rows(h, s, n, bs, block0, seed, stream_id, 1, stream))) return rc;
    a.prior = s->bp_rows; a.prior_stride = 0;
  }
  a.dw = pl.b->dw; a.n_calls = 1; a.x = x; a.y = y; a.v = v; a.z = const_cast<float *>(z); a.n = n; a.row_base = 0;
  a.bs = bs; a.block0 = block0; a.tiles_per_block = tpb; a.n_items = n_blocks * tpb; a.mode = 0;
  a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.stream0 = stream_id; a.out = out; a.ws = pl.b->ws; a.ws_stride = pl.ws_stride;
  hipLaunchKernelGGL(bnw_rows_kernel, dim3(pl.grid), dim3(BNN_THREADS), BNW_LDS_BYTES, stream, a);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

namespace {
int bnw_effects_of(const BnwNets &mf, const BnwPlan &pl, float *dw_eff, const float *z, long long n, int bs, int n_blocks, int block0, long long row_base,
                   int n_doses, const float *xvals, uint64_t seed, uint32_t stream0, uint32_t it_noise, int sample_y, double *sum_out,
                   long long sum_stride, float *ite_out, long long ite_stride, hipStream_t stream, const double *stats = nullptr) {
  bnw_noise(mf, dw_eff, n_blocks, n_doses, block0, seed, stream0, 1u, stream);
  BnwEffArgs e{};
  e.stats = stats;
  e.m = mf; e.mp = mf.dev; e.dw = dw_eff; e.z = z; e.n = n; e.row_base = row_base; e.bs = bs; e.block0 = block0;
  e.tiles_per_block = (bs + BNW_RT - 1) / BNW_RT; e.n_items = n_blocks * e.tiles_per_block; e.n_doses = n_doses; e.xvals = xvals;
  e.k0 = (uint32_t)seed; e.k1 = (uint32_t)(seed >> 32); e.stream0 = stream0; e.it_noise = it_noise; e.sample_y = sample_y;
  e.sum_out = sum_out; e.sum_stride = sum_stride; e.ite_out = ite_out; e.ite_stride = ite_stride;
  e.ws = pl.b->ws; e.ws_stride = pl.ws_stride;
  hipLaunchKernelGGL(bnw_effects_kernel, dim3(std::max(1, std::min(e.n_items, pl.grid))), dim3(BNN_THREADS), BNW_LDS_BYTES, stream, e);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}
}  // namespace

int bnw_mh_run(bgm_handle *h, BnnState *s, const bgm_bnn_mh_args *g, hipStream_t stream) {
  int rc = bnw_need(s, "bgm_bnn_mh_run");
  if (rc) return rc;
  const int ids[3] = {BNN_G, BNN_H, BNN_F}, idf[1] = {BNN_F};
  BnwRowsArgs a{};
  bnw_nets(s, ids, 3, a.m);
  BnwNets mf;
  bnw_nets(s, idf, 1, mf);
  const long long n = g->n;
  const int bs = g->block_rows, n_blocks = (int)((n + bs - 1) / bs), tpb = (bs + BNW_RT - 1) / BNW_RT, q = s->q;
  const int n_doses = g->effect == 1 ? g->n_doses : (g->effect == 2 ? 2 : 0);
  BnwPlan pl;
  // one buffer: [2 sets per block of g | h | f] then [n_doses sets per block of f]
  const size_t mh_floats = (size_t)2 * n_blocks * a.m.set_floats, eff_floats = (size_t)n_blocks * n_doses * mf.set_floats;
  rc = bnw_plan(h, s, a.m, (long long)((mh_floats + eff_floats) / std::max<long long>(1, a.m.set_floats) + 2), n_blocks * tpb, pl, stream);
  if (rc) return rc;
  if ((rc = bnw_pack(pl.b, a.m, 0, stream)) || (rc = bnw_pack(pl.b, mf, ((size_t)a.m.set_floats + 63) & ~(size_t)63, stream))) return rc;
  a.mp = a.m.dev;
  float *dw_eff = pl.b->dw + ((mh_floats + 63) & ~(size_t)63);
  a.dw = pl.b->dw; a.n_calls = 2; a.x = g->x_dev; a.y = g->y_dev; a.v = g->v_dev; a.z = g->state_dev; a.n = n; a.row_base = g->row_base;
  a.bs = bs; a.block0 = g->block0; a.tiles_per_block = tpb; a.n_items = n_blocks * tpb; a.mode = 1;
  a.q_sd = g->q_sd; a.q_sd_blocks = g->q_sd_blocks_dev;
  a.k0 = (uint32_t)g->seed; a.k1 = (uint32_t)(g->seed >> 32); a.acc_count = g->acc_count_dev;
  a.ws = pl.b->ws; a.ws_stride = pl.ws_stride;
  BnwBatch bt;
  rc = bnw_batch(s, pl.b, n_blocks, bt, stream);
  if (rc) return rc;
  a.xstats = bt.xstats;
  // effects of the draw kept by iteration it_kept; batch statistics: those of the states AFTER its accept step = slot 1 of the
  // statistics pass in front of the next iteration (parity `par`)
  auto effects = [&](int it_kept, int par) -> int {
    const int d = it_kept - g->burn_in;
    return bnw_effects_of(mf, pl, dw_eff, g->state_dev, n, bs, n_blocks, g->block0, g->row_base, n_doses,
                          g->effect == 1 ? g->x_values_dev : pl.b->pair, g->seed, 0x40000000u + (uint32_t)d * (uint32_t)n_doses, (uint32_t)it_kept,
                          g->sample_y, g->effect == 1 ? g->adrf_sum_dev + d : nullptr, g->n_keep, g->effect == 2 ? g->ite_dev + d : nullptr,
                          g->n_keep, stream, bt.on ? bt.parity(par) : nullptr);
  };
  auto kept = [&](int it) { return g->effect && it >= g->burn_in && it - g->burn_in < g->n_keep; };
  for (int i = 0; i < g->n_iters; ++i) {
    const int it = g->it_begin + i;
    bnw_noise(a.m, pl.b->dw, n_blocks, 2, g->block0, g->seed, 2u * (uint32_t)it, 1u, stream);
    a.it = it; a.init = (i == 0 && g->init) ? 1 : 0;
    if (bt.on) {
      bnw_stats(bt, g->state_dev, n, g->row_base, q, bs, it, a.init, g->q_sd, g->q_sd_blocks_dev, g->seed, i & 1, g->x_dev, i == 0, stream);
      if (i > 0 && kept(it - 1) && (rc = effects(it - 1, i & 1))) return rc;
      a.stats = bt.parity(i & 1);
    }
    if (s->bp_on) {                     // the two evaluations' own calls of the prior net (streams 2 it, 2 it + 1)
      if ((rc = bprior_rows(h, s, n, bs, g->block0, g->seed, 2u * (uint32_t)it, 2, stream))) return rc;
      a.prior = s->bp_rows; a.prior_stride = n * (long long)(q + 2);
    }
    a.acc_blocks = g->acc_blocks_dev ? g->acc_blocks_dev + (long long)i * n_blocks : nullptr;
    hipLaunchKernelGGL(bnw_rows_kernel, dim3(pl.grid), dim3(BNN_THREADS), BNW_LDS_BYTES, stream, a);
    const int d = it - g->burn_in;
    if (d >= 0 && d < g->n_keep) {
      if (g->draws_dev)
        BGM_HIP_CHECK(hipMemcpyAsync(g->draws_dev + (long long)d * n * q, g->state_dev, sizeof(float) * n * q, hipMemcpyDeviceToDevice, stream));
      if (!bt.on && g->effect && (rc = effects(it, 0))) return rc;
    }
  }
  if (bt.on && g->n_iters > 0 && kept(g->it_begin + g->n_iters - 1)) {      // statistics of the final states: one more pass (its proposals are not used)
    const int i = g->n_iters;
    bnw_stats(bt, g->state_dev, n, g->row_base, q, bs, g->it_begin + i, 0, g->q_sd, g->q_sd_blocks_dev, g->seed, i & 1, g->x_dev, false, stream);
    if ((rc = effects(g->it_begin + i - 1, i & 1))) return rc;
  }
  BGM_HIP_CHECK(hipGetLastError());
#ifdef BNW_PROF
  {
    unsigned long long acc[8], zero[8] = {0};
    BGM_HIP_CHECK(hipStreamSynchronize(stream));
    BGM_HIP_CHECK(hipMemcpyFromSymbol(acc, HIP_SYMBOL(bnw_prof_acc), sizeof(acc)));
    BGM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(bnw_prof_acc), zero, sizeof(zero)));
    const double tot = (double)(acc[0] + acc[1] + acc[2]);
    fprintf(stderr, "BNW_PROF net calls: signs %.3f  input normalisation %.3f  layers %.3f  (of %.3e cycles inside calls)\n", acc[0] / tot, acc[1] / tot, acc[2] / tot, tot);
    fprintf(stderr, "BNW_PROF layer products: first fetch %.3f  chunk loops %.3f  epilogues %.3f of the calls' cycles; %.0f cycles per chunk of 16 K, %.0f per epilogue, %.1f chunks per pass\n",
            acc[3] / tot, acc[4] / tot, acc[5] / tot, (double)acc[4] / (double)acc[6], (double)acc[5] / (double)acc[7], (double)acc[6] / (double)acc[7]);
  }
#endif
  return BGM_OK;
}

int bnw_effects(bgm_handle *h, BnnState *s, const float *draws, int64_t n, int32_t bs, int32_t block0, int64_t row_base, int32_t n_keep, int32_t it0,
                uint64_t seed, int32_t effect, int32_t sample_y, const float *x_values, int32_t n_doses, double *adrf_sum, float *ite,
                hipStream_t stream) {
  int rc = bnw_need(s, "bgm_bnn_effects");
  if (rc) return rc;
  const int idf[1] = {BNN_F};
  BnwNets mf;
  bnw_nets(s, idf, 1, mf);
  const int n_blocks = (int)((n + bs - 1) / bs), tpb = (bs + BNW_RT - 1) / BNW_RT, nd = effect == 1 ? n_doses : 2, q = s->q;
  BnwPlan pl;
  rc = bnw_plan(h, s, mf, (long long)n_blocks * nd, n_blocks * tpb, pl, stream);
  if (rc) return rc;
  if ((rc = bnw_pack(pl.b, mf, 0, stream))) return rc;
  BnwBatch bt;
  rc = bnw_batch(s, pl.b, n_blocks, bt, stream);
  if (rc) return rc;
  for (int d = 0; d < n_keep; ++d) {
    const float *zd = draws + (long long)d * n * q;
    if (bt.on) bnw_stats(bt, zd, n, row_base, q, bs, d, 0, 0.0f, nullptr, seed, d & 1, nullptr, false, stream);      // statistics of this draw (slot 1)
    rc = bnw_effects_of(mf, pl, pl.b->dw, zd, n, bs, n_blocks, block0, row_base, nd, effect == 1 ? x_values : pl.b->pair, seed,
                        0x40000000u + (uint32_t)d * (uint32_t)nd, (uint32_t)(it0 + d), sample_y, effect == 1 ? adrf_sum + d : nullptr, n_keep,
                        effect == 2 ? ite + d : nullptr, n_keep, stream, bt.on ? bt.parity(d & 1) : nullptr);
    if (rc) return rc;
  }
  return BGM_OK;
}

int bnw_evaluate(bgm_handle *h, BnnState *s, const float *x, const float *y, const float *v, float *z, int32_t encode, int64_t n, const float *x_values,
                 int32_t n_doses, uint64_t seed, uint32_t stream_id, double *sums, double *dose_sums, float *ite, hipStream_t stream) {
  int rc = bnw_need(s, "bgm_bnn_evaluate");
  if (rc) return rc;
  const int ids[4] = {BNN_G, BNN_H, BNN_F, BNN_E}, idf[1] = {BNN_F};
  BnwRowsArgs a{};
  bnw_nets(s, ids, 4, a.m);
  BnwNets mf;
  bnw_nets(s, idf, 1, mf);
  const int nd = dose_sums ? n_doses : (ite ? 2 : 0);
  const int bs = (int)n, tpb = (bs + BNW_RT - 1) / BNW_RT;        // the whole panel is ONE block (base.py:534-570)
  BnwPlan pl;
  const size_t all_floats = (size_t)a.m.set_floats, eff_floats = (size_t)nd * mf.set_floats;
  rc = bnw_plan(h, s, a.m, (long long)((all_floats + eff_floats) / std::max<long long>(1, a.m.set_floats) + 2), tpb, pl, stream);
  if (rc) return rc;
  if ((rc = bnw_pack(pl.b, a.m, 0, stream)) || (rc = bnw_pack(pl.b, mf, ((size_t)a.m.set_floats + 63) & ~(size_t)63, stream))) return rc;
  a.mp = a.m.dev;
  float *dw_eff = pl.b->dw + ((all_floats + 63) & ~(size_t)63);
  bnw_noise(a.m, pl.b->dw, 1, 1, 0, seed, stream_id, 0u, stream);
  a.dw = pl.b->dw; a.n_calls = 1; a.x = x; a.y = y; a.v = v; a.z = z; a.n = n; a.row_base = 0; a.bs = bs; a.block0 = 0;
  a.tiles_per_block = tpb; a.n_items = tpb; a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.stream0 = stream_id;
  a.ws = pl.b->ws; a.ws_stride = pl.ws_stride;
  BnwBatch bt;                          // "batch": the whole panel is the batch of every call
  rc = bnw_batch(s, pl.b, 1, bt, stream);
  if (rc) return rc;
  if (encode) {
    if (bt.on) {
      hipLaunchKernelGGL(bnw_colstats_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 256), (unsigned)std::min(s->p, 1024)), dim3(256), 0, stream,
                         v, (long long)n, s->p, bt.vstats);
      a.vstats = bt.vstats;
    }
    a.mode = 3;
    hipLaunchKernelGGL(bnw_rows_kernel, dim3(pl.grid), dim3(BNN_THREADS), BNW_LDS_BYTES, stream, a);
    a.vstats = nullptr;
  }
  if (bt.on && (sums || nd)) {          // statistics of z (slot 1) and of x
    bnw_stats(bt, z, n, 0, s->q, bs, 0, 0, 0.0f, nullptr, seed, 0, x, x != nullptr, stream);
    a.stats = bt.parity(0); a.xstats = bt.xstats;
  }
  if (sums) {
    BGM_HIP_CHECK(hipMemsetAsync(sums, 0, 3 * sizeof(double), stream));
    a.mode = 2; a.sums = sums;
    hipLaunchKernelGGL(bnw_rows_kernel, dim3(pl.grid), dim3(BNN_THREADS), BNW_LDS_BYTES, stream, a);
  }
  if (nd) {
    if (dose_sums) BGM_HIP_CHECK(hipMemsetAsync(dose_sums, 0, sizeof(double) * nd, stream));
    rc = bnw_effects_of(mf, pl, dw_eff, z, n, bs, 1, 0, 0, nd, dose_sums ? x_values : pl.b->pair, seed, stream_id + 1u, 0u, 0, dose_sums, 1,
                        dose_sums ? nullptr : ite, 1, stream, bt.on ? bt.parity(0) : nullptr);
    if (rc) return rc;
  }
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}
