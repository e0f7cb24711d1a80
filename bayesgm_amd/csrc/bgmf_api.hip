// bgmf_api.hip -- host side of the register-chained frozen-noise HMC of BGM with the Bayesian generator (bgmf_kernels.h):
// blob layout, eligibility, pack + launch.  Called from bgm_bvn_hmc_run (bgmb_api.hip).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "bgm_host.h"
#include "bgmb_state.h"
#include "bgmfx_kernels.h"

struct BgmfState {
  BgmfMeta m{};
  BgmfPackArgs pk{};
  float *blob = nullptr;       // frozen noise: resident part + one slot of chunks
  float *blob_fresh = nullptr; // fresh noise: resident part + fresh_slots slots [dW first layer | dW hidden | head blocks]
  int fresh_slots = 0;
  int lds_bytes = 0, ktq = 1;
  // split precision (bgmfx_kernels.h): the stream of fp16 fragments, the fp32 resident part, their meta (LDS offsets); 0 bytes = not served
  BgmfMeta xm{};
  unsigned char *sx = nullptr;
  float *xres = nullptr;
  int lds_bytes_x = 0;
};

void bgmf_free(BgmbState *s) {
  BgmfState *f = static_cast<BgmfState *>(s->bgmf);
  if (!f) return;
  if (f->blob) hipFree(f->blob);
  if (f->blob_fresh) hipFree(f->blob_fresh);
  if (f->sx) hipFree(f->sx);
  if (f->xres) hipFree(f->xres);
  delete f;
  s->bgmf = nullptr;
}

// 1: this generator is outside the compiled chains (hidden widths other than 64, depth, latent width, LDS): the LDS-tile engine serves it
static int bgmf_session(BgmbState *s, BgmfState *&out) {
  out = static_cast<BgmfState *>(s->bgmf);
  if (out) return out->blob ? BGM_OK : 1;
  const BnnNet &n = s->net;
  const int NF = n.n_layers, nh = NF - 2, q = s->q, p = s->p;
  BgmfState *f = new BgmfState();
  s->bgmf = f;                                             // (kept also when not eligible: the answer does not change for this session)
  bool ok = (nh == 3 || nh == 5) && q >= 1 && q <= 32 && p >= 1 && !std::getenv("BGM_BVN_NO_CHAINS");
  for (int l = 0; l < nh && ok; ++l) ok = n.lout[l] == 64;
  if (!ok) return 1;
  const int ktq = (q + 15) / 16, ntx = (p + 15) / 16;
  BgmfMeta &m = f->m;
  m.q = q; m.p = p; m.nh = nh; m.ntx = ntx;
  int off = 0;
  auto take = [&](int cnt) { const int o = off; off += (cnt + 3) / 4 * 4; return o; };
  m.w1 = take(4 * 16 * ktq * 17); m.d1 = take(4 * 16 * ktq * 17); m.b1 = take(64);
  m.wh = take((nh - 1) * 4 * 64 * 17); m.bh = take((nh - 1) * 64);
  m.bhd = take(2 * 16 * ntx);
  m.sc = take(16 * ktq); m.sh = take(16 * ktq);
  m.resident = off;
  m.chunks = off; off += (nh - 1 + ntx) * BGMF_CHUNK;
  m.total = off;
  m.slot_floats = (nh + ntx) * BGMF_CHUNK;
  for (int l = 0; l < NF; ++l) { m.sin_w[l] = n.sin_w[l]; m.sout_w[l] = n.sout_w[l]; f->pk.woff[l] = n.woff[l]; f->pk.eoff[l] = n.eoff[l]; }
  m.swords = (n.swords + 3) / 4 * 4;
  m.swp = m.swords | 1;
  m.stage = m.resident;
  m.sign = m.stage + 2 * BGMF_CHUNK;
  f->lds_bytes = (int)sizeof(float) * (m.sign + 16 * BGMF_WAVES * m.swp);
  f->ktq = ktq;
  if (f->lds_bytes > 160 * 1024) return 1;
  if (hipMalloc((void **)&f->blob, sizeof(float) * (size_t)m.total) != hipSuccess || hipMemset(f->blob, 0, sizeof(float) * (size_t)m.total) != hipSuccess) {
    f->blob = nullptr;
    bgm_set_error("frozen-noise HMC: device allocation failed");
    return BGM_E_HIP;
  }
  f->pk.m = m; f->pk.blob = f->blob; f->pk.ktq = ktq;
  out = f;
  return BGM_OK;
}

// Split precision: meta and buffers of the streamed form, made on first use.  0: ready; 1: not served (z_dim > 16, LDS); < 0: error
static int bgmfx_session(BgmfState *f) {
  if (f->sx) return BGM_OK;
  if (f->ktq != 1) return 1;
  BgmfMeta &x = f->xm;
  x = f->m;
  int off = 0;
  auto take = [&](int cnt) { const int o = off; off += (cnt + 3) / 4 * 4; return o; };
  x.w1 = x.d1 = x.wh = 0;
  x.b1 = take(64); x.bh = take((x.nh - 1) * 64); x.bhd = take(2 * 16 * x.ntx); x.sc = take(16); x.sh = take(16);
  x.resident = off;
  x.stage = off; off += 2 * BGM_X3_STEP * (BGM_X3_BLOCK_BYTES / 4);
  x.sign = off;
  f->lds_bytes_x = (int)sizeof(float) * (x.sign + 16 * BGMFX_WAVES * x.swp);
  if (f->lds_bytes_x > 160 * 1024) { f->lds_bytes_x = 0; return 1; }
  const size_t sx_bytes = (size_t)(2 * x.nh - 1 + x.ntx) * BGM_X3_STEP * BGM_X3_BLOCK_BYTES;
  if (hipMalloc((void **)&f->sx, sx_bytes) != hipSuccess || hipMemset(f->sx, 0, sx_bytes) != hipSuccess ||
      hipMalloc((void **)&f->xres, sizeof(float) * (size_t)x.resident) != hipSuccess || hipMemset(f->xres, 0, sizeof(float) * (size_t)x.resident) != hipSuccess) {
    if (f->sx) hipFree(f->sx);
    if (f->xres) hipFree(f->xres);
    f->sx = nullptr; f->xres = nullptr;
    bgm_set_error("frozen-noise HMC (split precision): device allocation failed");
    return BGM_E_HIP;
  }
  return BGM_OK;
}

// bgm_bvn_set_precision: 0 fp32 | 2 f16x3 (frozen noise on the reference's generator shape with z_dim <= 16 only)
int bgmf_set_precision(BgmbState *s, int mode) {
  if (mode != 0 && mode != 2) { bgm_set_error("bgm_bvn_set_precision: mode 0 (fp32) or 2 (f16x3)"); return BGM_E_INVALID; }
  if (mode == 2) {
    BgmfState *f;
    int rc = s->cfg.hmc_frozen_noise ? bgmf_session(s, f) : 1;
    if (rc == 0) rc = bgmfx_session(f);
    if (rc < 0) return rc;
    if (rc) { bgm_set_error("bgm_bvn_set_precision: f16x3 serves frozen-noise HMC of generators with 3 or 5 hidden layers of 64 units and z_dim <= 16"); return BGM_E_UNSUPPORTED; }
  }
  s->precision = mode;
  return BGM_OK;
}

static int bgmfx_hmc(bgm_handle *h, BgmbState *s, BgmfState *f, const bgm_hmc_args *g, hipStream_t st) {
  int rc = bgmfx_session(f);
  if (rc) return rc < 0 ? rc : BGM_E_UNSUPPORTED;
  BgmfxPackArgs pk{};
  pk.m = f->xm;
  for (int l = 0; l < BGMF_MAXNH + 2; ++l) { pk.woff[l] = f->pk.woff[l]; pk.eoff[l] = f->pk.eoff[l]; }
  pk.theta = s->theta_dev; pk.dwc = s->dw_dev; pk.bnp = s->theta_dev + s->net.off; pk.sx = f->sx; pk.res = f->xres;
  const int n_steps = 2 * f->xm.nh - 1 + f->xm.ntx;
  hipLaunchKernelGGL(bgmfx_pack_kernel, dim3((unsigned)n_steps + 1, 16), dim3(512), 0, st, pk);      // fragments of the CURRENT parameters and of this run's perturbation
  BGM_HIP_CHECK(hipGetLastError());
  BgmfxHmcKArgs x{};
  BgmfHmcKArgs &k = x.k;
  k.blob = f->xres; k.x = g->x_dev; k.n = g->n; k.row_base = g->row_base; k.state = g->state_dev; k.logp = g->logp_dev; k.grad = g->grad_dev;
  k.init = g->init; k.it_begin = g->it_begin; k.n_iters = g->n_iters; k.burn_in = g->burn_in; k.n_leapfrog = g->n_leapfrog; k.step = g->step_dev;
  k.k0 = (uint32_t)(g->seed & 0xFFFFFFFFull); k.k1 = (uint32_t)(g->seed >> 32);
  k.acc_prob_sum = g->acc_prob_sum_dev; k.acc_count = g->acc_count_dev; k.draws = g->draws_dev;
  k.m = f->xm;
  x.sx = f->sx;
  const long long tiles = (g->n + 15) / 16;
  const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((tiles + BGMFX_WAVES - 1) / BGMFX_WAVES, h->n_cus));
  auto launch = [&](auto kern) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, f->lds_bytes_x) != hipSuccess) return BGM_E_HIP;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * BGMFX_WAVES), f->lds_bytes_x, st, x);
    return hipGetLastError() == hipSuccess ? BGM_OK : BGM_E_HIP;
  };
  if ((f->xm.p & 3) == 0) rc = f->xm.nh == 5 ? launch(bgmfx_hmc_kernel<5, BGMFX_WAVES, true>) : launch(bgmfx_hmc_kernel<3, BGMFX_WAVES, true>);
  else rc = f->xm.nh == 5 ? launch(bgmfx_hmc_kernel<5, BGMFX_WAVES, false>) : launch(bgmfx_hmc_kernel<3, BGMFX_WAVES, false>);
  if (rc) bgm_set_error("frozen-noise HMC (bgmfx_hmc_kernel): launch failed");
  return rc;
}

// 0: launched; 1: not this kernel's shape; < 0: error
int bgmf_hmc_try(bgm_handle *h, BgmbState *s, const bgm_hmc_args *g, hipStream_t st) {
  BgmfState *f;
  int rc = bgmf_session(s, f);
  if (rc) return rc;
  if (s->precision == 2) return bgmfx_hmc(h, s, f, g, st);
  BgmfPackArgs pk = f->pk;
  pk.theta = s->theta_dev; pk.dwc = s->dw_dev; pk.bnp = s->theta_dev + s->net.off; pk.fresh = 0; pk.dw_stride = 0;
  hipLaunchKernelGGL(bgmf_pack_kernel, dim3(32, f->m.nh + 1, 1), dim3(256), 0, st, pk);        // blob of the CURRENT parameters and of this run's perturbation
  BGM_HIP_CHECK(hipGetLastError());
  BgmfHmcKArgs k{};
  k.blob = f->blob; k.x = g->x_dev; k.n = g->n; k.row_base = g->row_base; k.state = g->state_dev; k.logp = g->logp_dev; k.grad = g->grad_dev;
  k.init = g->init; k.it_begin = g->it_begin; k.n_iters = g->n_iters; k.burn_in = g->burn_in; k.n_leapfrog = g->n_leapfrog; k.step = g->step_dev;
  k.k0 = (uint32_t)(g->seed & 0xFFFFFFFFull); k.k1 = (uint32_t)(g->seed >> 32);
  k.acc_prob_sum = g->acc_prob_sum_dev; k.acc_count = g->acc_count_dev; k.draws = g->draws_dev;
  k.m = f->m;
  const long long tiles = (g->n + 15) / 16;
  const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((tiles + BGMF_WAVES - 1) / BGMF_WAVES, h->n_cus));
  auto launch = [&](auto kern) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, f->lds_bytes) != hipSuccess) return BGM_E_HIP;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * BGMF_WAVES), f->lds_bytes, st, k);
    return hipGetLastError() == hipSuccess ? BGM_OK : BGM_E_HIP;
  };
  if (f->ktq == 1) rc = f->m.nh == 5 ? launch(bgmf_hmc_kernel<1, 5, false>) : launch(bgmf_hmc_kernel<1, 3, false>);
  else rc = f->m.nh == 5 ? launch(bgmf_hmc_kernel<2, 5, false>) : launch(bgmf_hmc_kernel<2, 3, false>);
  if (rc) bgm_set_error("frozen-noise HMC (bgmf_hmc_kernel): launch failed");
  return rc;
}


// Fresh noise: one launch of transitions [it_begin, it_begin + n_iters) whose perturbations (n_iters * L slots, then the initial
// evaluation's when `init`) lie in s->dw_dev, dw_stride floats apart (bvn_noise).  0: launched; 1: not this kernel's shape; < 0: error
int bgmf_hmc_fresh(bgm_handle *h, BgmbState *s, const bgm_hmc_args *g, int it_begin, int n_iters, int init, long long dw_stride, hipStream_t st) {
  BgmfState *f;
  int rc = bgmf_session(s, f);
  if (rc) return rc;
  const int slots = n_iters * g->n_leapfrog + (init ? 1 : 0);
  if (slots > f->fresh_slots) {
    if (f->blob_fresh) { BGM_HIP_CHECK(hipDeviceSynchronize()); hipFree(f->blob_fresh); f->blob_fresh = nullptr; f->fresh_slots = 0; }
    const size_t floats = (size_t)f->m.chunks + (size_t)slots * (size_t)f->m.slot_floats;
    if (hipMalloc((void **)&f->blob_fresh, sizeof(float) * floats) != hipSuccess || hipMemset(f->blob_fresh, 0, sizeof(float) * floats) != hipSuccess) {
      f->blob_fresh = nullptr;
      bgm_set_error("fresh-noise HMC: device allocation failed");
      return BGM_E_HIP;
    }
    f->fresh_slots = slots;
  }
  BgmfPackArgs pk = f->pk;
  pk.blob = f->blob_fresh; pk.theta = s->theta_dev; pk.dwc = s->dw_dev; pk.bnp = s->theta_dev + s->net.off; pk.fresh = 1; pk.dw_stride = dw_stride;
  hipLaunchKernelGGL(bgmf_pack_kernel, dim3(16, f->m.nh + 1, (unsigned)slots), dim3(256), 0, st, pk);
  BGM_HIP_CHECK(hipGetLastError());
  BgmfHmcKArgs k{};
  k.blob = f->blob_fresh; k.x = g->x_dev; k.n = g->n; k.row_base = g->row_base; k.state = g->state_dev; k.logp = g->logp_dev; k.grad = g->grad_dev;
  k.init = init; k.it_begin = it_begin; k.n_iters = n_iters; k.burn_in = g->burn_in; k.n_leapfrog = g->n_leapfrog; k.step = g->step_dev;
  k.k0 = (uint32_t)(g->seed & 0xFFFFFFFFull); k.k1 = (uint32_t)(g->seed >> 32);
  k.acc_prob_sum = g->acc_prob_sum_dev; k.acc_count = g->acc_count_dev; k.draws = g->draws_dev;
  k.m = f->m;
  const long long tiles = (g->n + 15) / 16;
  const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((tiles + BGMF_WAVES - 1) / BGMF_WAVES, h->n_cus));
  auto launch = [&](auto kern) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, f->lds_bytes) != hipSuccess) return BGM_E_HIP;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * BGMF_WAVES), f->lds_bytes, st, k);
    return hipGetLastError() == hipSuccess ? BGM_OK : BGM_E_HIP;
  };
  if (f->ktq == 1) rc = f->m.nh == 5 ? launch(bgmf_hmc_kernel<1, 5, true>) : launch(bgmf_hmc_kernel<1, 3, true>);
  else rc = f->m.nh == 5 ? launch(bgmf_hmc_kernel<2, 5, true>) : launch(bgmf_hmc_kernel<2, 3, true>);
  if (rc) bgm_set_error("fresh-noise HMC (bgmf_hmc_kernel): launch failed");
  return rc;
}
