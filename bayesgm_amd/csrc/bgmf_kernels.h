// bgmf_kernels.h -- HMC of BGM with the Bayesian generator under frozen noise as register-chained row tiles on gfx950.
//
// replaces: BGM.tfp_mcmc_sampler bgm/base.py:709-830 on the target bgm/base.py:665-705 with g_net = BayesianVariationalNet
// (networks/bnn.py:40-99; tfp.layers.DenseFlipout restated in oracle/bnn.py, oracle/bgm_bnn.py hmc_sampler(frozen=True)) for the
// reference's generator shape: hidden layers of 64 units.  The execution scheme is the deterministic sampler's (bgm_kernels.h): a wave
// owns 16 chains, activations, the leapfrog state and dlogp/dz live in registers in the layout one layer's MFMA output is the next
// layer's B operand, weights are read from LDS in the dual-access [out tile][in row][17] layout.  A Flipout layer
//     y = h loc + ((h * s_in) dW) * s_out + b
// is two such products: the posterior means `loc` of the trunk are LDS-resident for the whole launch; the run's ONE perturbation
// dW = sigma * eps (frozen noise, DESIGN_HISTORY.md section 7b) does not fit beside them and is streamed from L2 through a double-buffered
// LDS stage, one 64 x 64 layer (or one 16-feature block of both heads, loc and dW) per step, fetched one step ahead and shared by
// the eight waves of the workgroup, which walk the layers in lock step (one barrier per step).  The rows' sign strings are drawn once
// per row tile (Philox, oracle/bnn.py draw_noise: word w of row r = Philox(ctr = (r, w >> 2, stream 0, TAG_SIGN))[w & 3]) into LDS,
// the trunk's into 16-bit register masks.  Other generator shapes run on the LDS-tile engine (gx_flipout.h).
#pragma once
#include "bgm_kernels.h"
#include <type_traits>

#include "bnn_kernels.h"

#define BGMF_CHUNK (2 * BGM_PAIR)      // floats per stream step: one hidden dW layer [4][64][17], or [loc mean | loc var | dW mean | dW var] of a head block
#define BGMF_WAVES 8
#define BGMF_MAXNH 6

struct BgmfMeta {
  int q, p, nh, ntx;                   // latent dim, data dim, hidden layers of the trunk (all 64 wide), 16-feature blocks of the data dim
  int w1, d1, b1;                      // first layer: loc and dW [4][16 KTQ][17] (slot-permuted like BgmMeta::w1), bias [64]
  int wh, bh;                          // (nh - 1) x loc [4][64][17], (nh - 1) x [64]
  int bhd;                             // head biases [2][16 ntx]
  int sc, sh;                          // input BatchNorm (inference mode) as z * sc + sh, [16 KTQ] by feature, zero beyond q
  int resident;                        // floats copied to LDS at kernel start (everything above)
  int chunks;                          // blob offset of the stream: dW of hidden layers 1 .. nh-1, then the head blocks; BGMF_CHUNK floats each
                                       // fresh noise: per gradient evaluation (slot) [dW of the first layer | hidden 1 .. nh-1 | head blocks]
  int slot_floats;                     // fresh noise: floats of one slot = (nh + ntx) chunks
  int total;
  int stage, sign;                     // LDS offsets (floats): [2][BGMF_CHUNK], sign words [16 BGMF_WAVES][swp]
  int swp, swords;                     // row stride of the sign words in LDS (odd: conflict-free across rows), words drawn per row (multiple of 4)
  int sin_w[BGMF_MAXNH + 2], sout_w[BGMF_MAXNH + 2];   // word offsets of the layers' input / output sign vectors: trunk 0 .. nh-1, mean head, variance head
};

// The workgroup's position in the cyclic stream  dW_1 .. dW_{nh-1}, H_0 .. H_{ntx-1}, dW_{nh-1} .. dW_2  (then dW_1 again: it serves the
// last backward layer of one gradient evaluation and the first forward layer of the next).
struct BgmfStream {
  const float *src;
  float *buf;
  int cur, tid, pos, nhh, ntx, cycle;
  f32x4 r0, r1, r2;
  __device__ __forceinline__ int chunk_of(int k) const {
    return k < nhh ? k : (k < nhh + ntx ? k : nhh - 1 - (k - nhh - ntx));
  }
  __device__ __forceinline__ void fetch_next() {
    const int nx = pos + 1 == cycle ? 0 : pos + 1;
    const f32x4 *s = reinterpret_cast<const f32x4 *>(src + (long long)chunk_of(nx) * BGMF_CHUNK);
    r0 = s[tid];
    r1 = s[tid + 64 * BGMF_WAVES];
    if (tid + 2 * 64 * BGMF_WAVES < BGMF_CHUNK / 4) r2 = s[tid + 2 * 64 * BGMF_WAVES];
  }
  __device__ __forceinline__ void commit() {
    f32x4 *d = reinterpret_cast<f32x4 *>(buf + (cur ^ 1) * BGMF_CHUNK);
    d[tid] = r0;
    d[tid + 64 * BGMF_WAVES] = r1;
    if (tid + 2 * 64 * BGMF_WAVES < BGMF_CHUNK / 4) d[tid + 2 * 64 * BGMF_WAVES] = r2;
    __syncthreads();
    cur ^= 1;
    pos = pos + 1 == cycle ? 0 : pos + 1;
  }
  __device__ __forceinline__ const float *tile() const { return buf + cur * BGMF_CHUNK; }
  __device__ __forceinline__ int cycle_len() const { return cycle; }
  __device__ __forceinline__ void begin(const float *blob, const BgmfMeta &m, float *lds) {
    src = blob + m.chunks; buf = lds + m.stage; tid = threadIdx.x;
    nhh = m.nh - 1; ntx = m.ntx; cycle = nhh + ntx + (nhh - 1);
    cur = 1; pos = cycle - 1;
    fetch_next();
    commit();                          // dW_1 is current on entry to every gradient evaluation
  }
};

// Fresh noise (params['bnn_mcmc_noise'] = 'fresh', the reference as written: a new perturbation at every gradient evaluation): the
// stream is linear.  Evaluation e of a pass reads slot(e); its steps are  dW_0, dW_1 .. dW_{nh-1}, H_0 .. H_{ntx-1}, dW_{nh-1} .. dW_1, dW_0
// (the first layer's perturbation is a chunk too); the step after an evaluation's last one is the first of the next evaluation, after
// the pass's last evaluation the first of the pass (every pass of the block replays the same slots).
struct BgmfStreamFresh {
  const float *src;
  float *buf;
  int cur, tid, pos, nhh, ntx, steps, ev, n_ev, slot_floats, init_first;
  f32x4 r0, r1, r2;
  __device__ __forceinline__ int slot_of(int e) const { return init_first ? (e == 0 ? n_ev - 1 : e - 1) : e; }      // the initial evaluation's slot is the last one
  __device__ __forceinline__ int chunk_of(int k) const {
    if (k == 0 || k == steps - 1) return 0;
    if (k <= nhh + ntx) return k;
    return nhh - (k - nhh - ntx - 1);
  }
  __device__ __forceinline__ void fetch_next() {
    int nk = pos + 1, ne = ev;
    if (nk == steps) { nk = 0; ne = ev + 1 == n_ev ? 0 : ev + 1; }
    const f32x4 *s = reinterpret_cast<const f32x4 *>(src + (long long)slot_of(ne) * slot_floats + (long long)chunk_of(nk) * BGMF_CHUNK);
    r0 = s[tid];
    r1 = s[tid + 64 * BGMF_WAVES];
    if (tid + 2 * 64 * BGMF_WAVES < BGMF_CHUNK / 4) r2 = s[tid + 2 * 64 * BGMF_WAVES];
  }
  __device__ __forceinline__ void commit() {
    f32x4 *d = reinterpret_cast<f32x4 *>(buf + (cur ^ 1) * BGMF_CHUNK);
    d[tid] = r0;
    d[tid + 64 * BGMF_WAVES] = r1;
    if (tid + 2 * 64 * BGMF_WAVES < BGMF_CHUNK / 4) d[tid + 2 * 64 * BGMF_WAVES] = r2;
    __syncthreads();
    cur ^= 1;
    if (++pos == steps) { pos = 0; ev = ev + 1 == n_ev ? 0 : ev + 1; }
  }
  __device__ __forceinline__ const float *tile() const { return buf + cur * BGMF_CHUNK; }
  __device__ __forceinline__ int cycle_len() const { return steps; }
  __device__ __forceinline__ void begin(const float *blob, const BgmfMeta &m, float *lds, int n_evals, int init) {
    src = blob + m.chunks; buf = lds + m.stage; tid = threadIdx.x;
    nhh = m.nh - 1; ntx = m.ntx; steps = 2 * nhh + ntx + 2; slot_floats = m.slot_floats;
    n_ev = n_evals; init_first = init;
    cur = 1; pos = steps - 1; ev = n_ev - 1;
    fetch_next();
    commit();                          // the first evaluation's first-layer perturbation is current
  }
};

__device__ __forceinline__ float bgmf_flip(float v, unsigned bit) {
  return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) ^ (bit << 31));
}
// The masks are loop invariants of the whole run: without this the compiler expands each of them into its 16 sign words once, outside the
// transition loop, and spills (12 masks x 16 registers).  Re-deriving the bit at every use costs two VALU operations in the MFMA shadow.
__device__ __forceinline__ unsigned bgmf_here(unsigned mask) { asm volatile("" : "+v"(mask)); return mask; }
// out = in * s for the lane's 4 KT values; bit 4t + r of `mask` set = sign -1
template <int KT>
__device__ __forceinline__ void bgmf_flip_tiles(const f32x4 (&in)[KT], unsigned mask, f32x4 (&out)[KT]) {
  mask = bgmf_here(mask);
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = in[t][r];        // (a scalar copy: hipcc 7.2 reads element 0 for __builtin_bit_cast of a vector-element lvalue)
      out[t][r] = bgmf_flip(v, (mask >> (4 * t + r)) & 1u);
    }
}
template <int KT>
__device__ __forceinline__ void bgmf_zero(f32x4 (&a)[KT]) {
#pragma unroll
  for (int t = 0; t < KT; ++t) a[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}

// the dW products of one head block: the mean tile reads h * s_in(mean head), the variance tile h * s_in(variance head); the flips are
// made on the way into the matrix pipe (two XORs per K-step) instead of holding two more copies of h
__device__ __forceinline__ void bgmf_heads_fwd(const float *wl, int j, int g, const f32x4 (&h)[4], unsigned in_m, unsigned in_v, f32x4 (&ms)[2]) {
  BGM_OPAQUE2(j, g);
  in_m = bgmf_here(in_m); in_v = bgmf_here(in_v);
  const float *base = wl + (4 * g) * 17 + j;
  float c0 = base[0], c1 = base[64 * 17], n0 = 0.0f, n1 = 0.0f;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int t = s >> 2, r = s & 3;
    if (s + 1 < 16) {
      const float *row = base + (16 * ((s + 1) >> 2) + ((s + 1) & 3)) * 17;
      n0 = row[0];
      n1 = row[64 * 17];
    }
    const float hv_ = h[t][r];
    ms[0] = BGM_MFMA(c0, bgmf_flip(hv_, (in_m >> s) & 1u), ms[0]);
    ms[1] = BGM_MFMA(c1, bgmf_flip(hv_, (in_v >> s) & 1u), ms[1]);
    BGM_NO_HOIST();
    c0 = n0; c1 = n1;
  }
}
// ... and backward: dhm[ti] += dWmean (dmu * s_out), dhv[ti] += dWvar (ds * s_out); the input signs are applied after the last block
__device__ __forceinline__ void bgmf_heads_bwd(const float *wl, int i, int g, const f32x4 (&dms)[2], f32x4 (&dhm)[4], f32x4 (&dhv)[4]) {
  BGM_OPAQUE2(i, g);
  const float *base = wl + i * 17 + 4 * g;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float am[4], av[4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) { am[ti] = base[16 * ti * 17 + r]; av[ti] = base[64 * 17 + 16 * ti * 17 + r]; }
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      dhm[ti] = BGM_MFMA(am[ti], dms[0][r], dhm[ti]);
      dhv[ti] = BGM_MFMA(av[ti], dms[1][r], dhv[ti]);
    }
    BGM_NO_HOIST();
  }
}

// The lane's sign masks of the trunk (bit 4t + r = the sign of the feature register r of tile t holds) for its row.
template <int NH>
struct BgmfSigns {
  unsigned in[NH], out[NH], in_m, in_v;
};
// 16 mask bits of a 64-wide sign vector starting at word w0 of the row: feature 16t + 4g + r = bit 16 (t & 1) + 4g + r of word t >> 1
__device__ __forceinline__ unsigned bgmf_mask64(const unsigned *sg, int w0, int g) {
  const unsigned a = sg[w0], b = sg[w0 + 1];
  return ((a >> (4 * g)) & 0xFu) | (((a >> (16 + 4 * g)) & 0xFu) << 4) | (((b >> (4 * g)) & 0xFu) << 8) | (((b >> (16 + 4 * g)) & 0xFu) << 12);
}
// Draw the sign words of the wave's 16 rows (noise stream `stream`) into its LDS rows and gather the lane's trunk masks.  BLOCK: all
// waves of the block call this together (block barrier inside); otherwise the wave orders its own LDS traffic: the rows are private to
// the wave, whose LDS instructions execute in issue order (fresh noise redraws the signs at every gradient evaluation, and a wave
// without a tile must not have to match a barrier there).
template <int KTQ, int NH, bool BLOCK>
__device__ __forceinline__ void bgmf_signs(const BgmfMeta &m, unsigned *sg_row, unsigned rowid, int g, unsigned k0, unsigned k1, unsigned stream,
                                           BgmfSigns<NH> &S) {
  if (!BLOCK) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the previous evaluation's reads of these rows have returned
  for (int c = g; c < (m.swords >> 2); c += 4) {
    const uint4 w4 = philox4x32_10(rowid, (unsigned)c, stream, BNN_TAG_SIGN, k0, k1);
    sg_row[4 * c] = w4.x; sg_row[4 * c + 1] = w4.y; sg_row[4 * c + 2] = w4.z; sg_row[4 * c + 3] = w4.w;
  }
  if (BLOCK) __syncthreads();
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // layer 0 reads z: feature 16t + 4r + g in register r of tile t (q <= 32: one word)
  const unsigned w = sg_row[m.sin_w[0]];
  unsigned m0 = 0u;
#pragma unroll
  for (int t = 0; t < KTQ; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) m0 |= ((w >> (16 * t + 4 * r + g)) & 1u) << (4 * t + r);
  S.in[0] = m0;
#pragma unroll
  for (int l = 1; l < NH; ++l) S.in[l] = bgmf_mask64(sg_row, m.sin_w[l], g);
#pragma unroll
  for (int l = 0; l < NH; ++l) S.out[l] = bgmf_mask64(sg_row, m.sout_w[l], g);
  S.in_m = bgmf_mask64(sg_row, m.sin_w[NH], g);
  S.in_v = bgmf_mask64(sg_row, m.sin_w[NH + 1], g);
}

// y = acc_loc + acc_dw * s_out, the LeakyReLU mask (bit 4t + r of sgn), h = lrelu(y)
__device__ __forceinline__ void bgmf_activate(const f32x4 (&al)[4], const f32x4 (&ad)[4], unsigned s_out, unsigned &sgn, f32x4 (&h)[4]) {
  s_out = bgmf_here(s_out);
  sgn = 0u;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = ad[t][r];
      const float v = al[t][r] + bgmf_flip(d, (s_out >> (4 * t + r)) & 1u);
      sgn |= (v > 0.0f) ? (1u << (4 * t + r)) : 0u;
      h[t][r] = lrelu(v);
    }
  asm volatile("" : "+v"(sgn));
}

// log p(z | x_obs) and dlogp/dz of the wave's 16 chains under the frozen perturbation.  Layouts as bgm_logp_grad (wide variant): z has
// feature 16t + 4r + g in register r of tile t; xrow = the (clamped) data row, NaN = missing.  Every wave of the block must call
// this the same number of times (the stream's barriers).
template <int KTQ, int NH, bool WANT_GRAD, bool FRESH, class Stream>
__device__ __forceinline__ void bgmf_logp_grad(const float *lds, const BgmfMeta &m, int j, int g, const f32x4 (&z)[KTQ], const float *xrow,
                                               Stream &st, const BgmfSigns<NH> &S, const unsigned *sg_row, float &logp, f32x4 (&grad)[KTQ]) {
  unsigned sgn[NH];
  f32x4 h[4];
  {
    f32x4 zin[KTQ], zs[KTQ], al[4], ad[4];
#pragma unroll
    for (int t = 0; t < KTQ; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * t + 4 * r + g;
        zin[t][r] = fmaf(z[t][r], lds[m.sc + f], lds[m.sh + f]);
      }
    bgmf_flip_tiles<KTQ>(zin, S.in[0], zs);
    if (FRESH) st.fetch_next();
    bias17<4>(lds + m.b1, g, al);
    fwd17<KTQ, 4>(lds + m.w1, j, g, zin, al);
    bgmf_zero<4>(ad);
    fwd17<KTQ, 4>(FRESH ? st.tile() : lds + m.d1, j, g, zs, ad);
    bgmf_activate(al, ad, S.out[0], sgn[0], h);
    if (FRESH) st.commit();
  }
#pragma unroll
  for (int l = 1; l < NH; ++l) {
    BGM_NO_HOIST();
    st.fetch_next();
    f32x4 hs[4], al[4], ad[4];
    bias17<4>(lds + m.bh + (l - 1) * 64, g, al);
    fwd17<4, 4>(lds + m.wh + (l - 1) * (4 * 64 * 17), j, g, h, al);
    bgmf_flip_tiles<4>(h, S.in[l], hs);
    bgmf_zero<4>(ad);
    fwd17<4, 4>(st.tile(), j, g, hs, ad);
    bgmf_activate(al, ad, S.out[l], sgn[l], h);
    st.commit();
  }
  // heads, one 16-feature block of both per stream step
  float nll = 0.0f;
  f32x4 dh[4], dhm[4], dhv[4];
  bgmf_zero<4>(dh); bgmf_zero<4>(dhm); bgmf_zero<4>(dhv);
  {
#pragma unroll 1
    for (int tx = 0; tx < m.ntx; ++tx) {
      BGM_NO_HOIST();
      st.fetch_next();
      const float *wl = st.tile();
      f32x4 ms[2], md[2];
      ms[0] = *reinterpret_cast<const f32x4 *>(lds + m.bhd + 16 * tx + 4 * g);
      ms[1] = *reinterpret_cast<const f32x4 *>(lds + m.bhd + 16 * (m.ntx + tx) + 4 * g);
      heads_fwd17(wl, j, g, h, ms);
      md[0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; md[1] = md[0];
      bgmf_heads_fwd(wl + BGM_PAIR, j, g, h, S.in_m, S.in_v, md);
      const int sh_ = 16 * (tx & 1) + 4 * g;
      const unsigned bm = (sg_row[m.sout_w[NH] + (tx >> 1)] >> sh_) & 0xFu, bv = (sg_row[m.sout_w[NH + 1] + (tx >> 1)] >> sh_) & 0xFu;
      f32x4 dms[2], dmf[2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * tx + 4 * g + r;
        const float x_ = (c < m.p) ? xrow[c] : 0.0f;
        const float d0 = md[0][r], d1 = md[1][r];
        const float mu = ms[0][r] + bgmf_flip(d0, (bm >> r) & 1u), sr = ms[1][r] + bgmf_flip(d1, (bv >> r) & 1u);
        const bool obs = (x_ == x_) && (c < m.p);   // NaN = missing
        const float s2 = softplus_f(sr) + BGM_EPS;
        const float inv = fast_rcp(s2);
        const float d = obs ? x_ - mu : 0.0f;
        nll += obs ? 0.5f * (d * d * inv + fast_log(s2)) : 0.0f;
        if (WANT_GRAD) {
          const float sg_ = fast_rcp(1.0f + fast_exp(-sr));
          const float gm = d * inv, gs = obs ? (0.5f * d * d * inv * inv - 0.5f * inv) * sg_ : 0.0f;
          dms[0][r] = gm; dms[1][r] = gs;
          dmf[0][r] = bgmf_flip(gm, (bm >> r) & 1u); dmf[1][r] = bgmf_flip(gs, (bv >> r) & 1u);
        }
      }
      if (WANT_GRAD) {
        heads_bwd17(wl, j, g, dms, dh);
        bgmf_heads_bwd(wl + BGM_PAIR, j, g, dmf, dhm, dhv);
      }
      st.commit();
    }
  }
  float zsq = 0.0f;
#pragma unroll
  for (int t = 0; t < KTQ; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) zsq = fmaf(z[t][r], z[t][r], zsq);
  logp = -sum_over_g(nll + 0.5f * zsq);
  if (WANT_GRAD) {
    {
      const unsigned im = bgmf_here(S.in_m), iv = bgmf_here(S.in_v);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = dhm[t][r], b = dhv[t][r];
          dh[t][r] += bgmf_flip(a, (im >> (4 * t + r)) & 1u) + bgmf_flip(b, (iv >> (4 * t + r)) & 1u);
        }
    }
#pragma unroll
    for (int l = NH - 1; l >= 1; --l) {
      BGM_NO_HOIST();
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[t][r] *= ((sgn[l] >> (4 * t + r)) & 1u) ? 1.0f : BGM_LEAK;
      if (FRESH || l > 1) st.fetch_next();      // (frozen noise: layer 1's dW stays current for the next evaluation's forward pass)
      f32x4 dn[4], dd[4], ds[4];
      bgmf_zero<4>(dn); bgmf_zero<4>(dd);
      bwd17<4, 4>(lds + m.wh + (l - 1) * (4 * 64 * 17), j, g, dh, dn);
      bgmf_flip_tiles<4>(dh, S.out[l], ds);
      bwd17<4, 4>(st.tile(), j, g, ds, dd);
      const unsigned il = bgmf_here(S.in[l]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = dd[t][r];
          dh[t][r] = dn[t][r] + bgmf_flip(d, (il >> (4 * t + r)) & 1u);
        }
      if (FRESH || l > 1) st.commit();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[t][r] *= ((sgn[0] >> (4 * t + r)) & 1u) ? 1.0f : BGM_LEAK;
    f32x4 ds[4], gd[KTQ];
    bgmf_zero<KTQ>(grad); bgmf_zero<KTQ>(gd);
    if (FRESH) st.fetch_next();        // (the next evaluation's first-layer perturbation)
    bwd17<KTQ, 4>(lds + m.w1, j, g, dh, grad);
    bgmf_flip_tiles<4>(dh, S.out[0], ds);
    bwd17<KTQ, 4>(FRESH ? st.tile() : lds + m.d1, j, g, ds, gd);
    if (FRESH) st.commit();
    const unsigned i0 = bgmf_here(S.in[0]);
#pragma unroll
    for (int t = 0; t < KTQ; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = gd[t][r];
        const float gz = grad[t][r] + bgmf_flip(d, (i0 >> (4 * t + r)) & 1u);
        grad[t][r] = fmaf(gz, lds[m.sc + 16 * t + 4 * r + g], -z[t][r]);   // through the input affine; prior -|z|^2 / 2
      }
  }
}

struct BgmfHmcKArgs {
  const float *blob;
  const float *x;            // [n x p], NaN = missing
  long long n, row_base;
  float *state, *logp, *grad;
  int init, it_begin, n_iters, burn_in, n_leapfrog;
  const float *step;
  unsigned k0, k1;
  double *acc_prob_sum;
  unsigned *acc_count;
  float *draws;
  BgmfMeta m;
};

// The transition logic of bgm_hmc_kernel on the Flipout target.  FRESH = false: one perturbation and one sign string per row for the
// whole run (stream 0).  FRESH = true (the reference as written; oracle/bgm_bnn.py hmc_sampler(frozen=False)): gradient evaluation l of
// transition `it` draws perturbation and signs from stream 1 + it * L + l (slot (it - it_begin) * L + l of the launch's noise), the
// initial evaluation from stream 0 (the launch's last slot).
template <int KTQ, int NH, bool FRESH>
__global__ __launch_bounds__(64 * BGMF_WAVES) void bgmf_hmc_kernel(BgmfHmcKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BgmfMeta &m = a.m;
  lds_fill(lds, a.blob, m.resident);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int evals = (a.init ? 1 : 0) + a.n_iters * a.n_leapfrog;
  using Stream = typename std::conditional<FRESH, BgmfStreamFresh, BgmfStream>::type;
  Stream st;
  if constexpr (FRESH) st.begin(a.blob, m, lds, evals, a.init ? 1 : 0);
  else st.begin(a.blob, m, lds);
  unsigned *sg_row = reinterpret_cast<unsigned *>(lds + m.sign) + (16 * wave + j) * m.swp;
  const long long n = a.n, n_tiles = (n + 15) / 16, passes = bgm_block_passes(n_tiles, BGMF_WAVES);
  const float eps = *a.step;
  for (long long ps = 0; ps < passes; ++ps) {
    // tiles dealt wave-major; a wave without a tile in a partly filled last round only keeps the stream moving (bgm_hmc_kernel)
    long long tile = (ps * BGMF_WAVES + wave) * gridDim.x + blockIdx.x;
    const bool tile_ok = tile < n_tiles;
    tile = tile_ok ? tile : n_tiles - 1;
    long long row = tile * 16 + j;
    const bool ok = tile_ok && row < n;
    row = row < n ? row : n - 1;
    const unsigned rowid = (unsigned)(a.row_base + row);
    const float *xrow = a.x + row * (long long)m.p;
    BgmfSigns<NH> S;
    if constexpr (!FRESH) {
      __syncthreads();         // (the previous pass's readers of the sign rows are done)
      bgmf_signs<KTQ, NH, true>(m, sg_row, rowid, g, a.k0, a.k1, 0u, S);
    }
    if (!tile_ok) {
      const int steps = st.cycle_len();
      for (int e = 0; e < evals; ++e)
        for (int k = 0; k < steps; ++k) { st.fetch_next(); st.commit(); }
      continue;
    }
    f32x4 z[KTQ], gr[KTQ];
    float lp;
    if (a.init) {   // initial_state ~ N(0,1)  (bgm/base.py:778), RNG tag 0
#pragma unroll
      for (int t = 0; t < KTQ; ++t) {
        const f32x4 e = box_muller4(philox4x32_10(rowid, 0u, (unsigned)(g + 4 * t), TAG_INIT, a.k0, a.k1));
#pragma unroll
        for (int r = 0; r < 4; ++r) z[t][r] = (16 * t + 4 * r + g < m.q) ? e[r] : 0.0f;
      }
      if constexpr (FRESH) bgmf_signs<KTQ, NH, false>(m, sg_row, rowid, g, a.k0, a.k1, 0u, S);
      bgmf_logp_grad<KTQ, NH, true, FRESH>(lds, m, j, g, z, xrow, st, S, sg_row, lp, gr);
    } else {
      bgm_load_z<KTQ>(a.state, m.q, row, g, z);
      bgm_load_z<KTQ>(a.grad, m.q, row, g, gr);
      lp = a.logp[row];
    }
    for (int it = a.it_begin; it < a.it_begin + a.n_iters; ++it) {
      BGM_NO_HOIST();
      f32x4 mom[KTQ], zc[KTQ], gc[KTQ];
      float ke0 = 0.0f;
#pragma unroll
      for (int t = 0; t < KTQ; ++t) {
        const f32x4 e = box_muller4(philox4x32_10(rowid, (unsigned)it, (unsigned)(g + 4 * t), TAG_MOM, a.k0, a.k1));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pm = (16 * t + 4 * r + g < m.q) ? e[r] : 0.0f;
          ke0 = fmaf(pm, pm, ke0);
          mom[t][r] = fmaf(0.5f * eps, gr[t][r], pm);
          zc[t][r] = z[t][r];
        }
      }
      ke0 = sum_over_g(ke0);
      float lpc = lp;
      for (int l = 0; l < a.n_leapfrog; ++l) {
        BGM_NO_HOIST();
#pragma unroll
        for (int t = 0; t < KTQ; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) zc[t][r] = fmaf(eps, mom[t][r], zc[t][r]);
        if constexpr (FRESH) bgmf_signs<KTQ, NH, false>(m, sg_row, rowid, g, a.k0, a.k1, 1u + (unsigned)it * (unsigned)a.n_leapfrog + (unsigned)l, S);
        bgmf_logp_grad<KTQ, NH, true, FRESH>(lds, m, j, g, zc, xrow, st, S, sg_row, lpc, gc);
        const float kick = (l < a.n_leapfrog - 1) ? eps : 0.5f * eps;
#pragma unroll
        for (int t = 0; t < KTQ; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) mom[t][r] = fmaf(kick, gc[t][r], mom[t][r]);
      }
      float ke1 = 0.0f;
#pragma unroll
      for (int t = 0; t < KTQ; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) ke1 = fmaf(mom[t][r], mom[t][r], ke1);
      ke1 = sum_over_g(ke1);
      float log_ratio = -((-lpc + 0.5f * ke1) - (-lp + 0.5f * ke0));
      log_ratio = (log_ratio == log_ratio && fabsf(log_ratio) != INFINITY) ? log_ratio : -INFINITY;
      const uint4 w4 = philox4x32_10(rowid, (unsigned)it >> 2, 0u, TAG_HACC, a.k0, a.k1);
      const unsigned w_ = (it & 2) ? ((it & 1) ? w4.w : w4.z) : ((it & 1) ? w4.y : w4.x);
      const float u = u01_open(w_);
      const bool acc = logf(u) < log_ratio;
#pragma unroll
      for (int t = 0; t < KTQ; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          z[t][r] = acc ? zc[t][r] : z[t][r];
          gr[t][r] = acc ? gc[t][r] : gr[t][r];
        }
      lp = acc ? lpc : lp;
      {
        float pa = (ok && g == 0) ? expf(fminf(log_ratio, 0.0f)) : 0.0f;
        for (int off = 8; off > 0; off >>= 1) pa += __shfl_xor(pa, off);
        const unsigned cnt = (unsigned)__popcll(__ballot(acc && ok && g == 0));
        if (lane == 0) {
          if (a.acc_prob_sum) atomicAdd(a.acc_prob_sum + it, (double)pa);
          if (a.acc_count) atomicAdd(a.acc_count + it, cnt);
        }
      }
      if (a.draws != nullptr && it >= a.burn_in && ok)
        bgm_store_z<KTQ>(a.draws + (long long)(it - a.burn_in) * n * m.q, m.q, row, g, z);
    }
    if (ok) {
      bgm_store_z<KTQ>(a.state, m.q, row, g, z);
      bgm_store_z<KTQ>(a.grad, m.q, row, g, gr);
      if (g == 0) a.logp[row] = lp;
    }
  }
}

// Blob of one BayesianVariationalNet from the session's parameter vector (gamma | beta | moving mean | moving variance, then per
// Flipout layer loc [in x out], rho [in x out], bias [out]; layers = trunk ..., mean head, variance head) and the run's perturbation
// dW in the parameter vector's own order (bgmb_noise_kernel, generator call 0).  grid (x, nh + 1): y = 0 first layer + affine,
// 1 .. nh-1 hidden layers, nh the heads.  The blob is zero-filled once at allocation; padding positions are never written.
struct BgmfPackArgs {
  BgmfMeta m;
  int woff[BGMF_MAXNH + 2], eoff[BGMF_MAXNH + 2];     // BnnNet::woff / eoff of the Flipout layers
  const float *bnp;            // gamma | beta | moving mean | moving variance, q each
  const float *theta, *dwc;    // dwc: perturbations in the parameter vector's order, one slot of dw_stride floats per generator call
  float *blob;
  int ktq, fresh;              // fresh: grid.z slots, each [dW first layer | dW hidden | head blocks]; else one slot, first-layer dW resident
  long long dw_stride;
};
static __global__ __launch_bounds__(256) void bgmf_pack_kernel(BgmfPackArgs a) {
  const BgmfMeta &m = a.m;
  const int sec = blockIdx.y, slot = blockIdx.z, q = m.q, p = m.p, nh = m.nh, nhh = nh - 1;
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
  const float *dwc = a.dwc + (long long)slot * a.dw_stride;
  float *chunks = a.blob + m.chunks + (long long)slot * m.slot_floats;        // (frozen noise: one slot, slot_floats unused)
  const int c_hid = a.fresh ? 1 : 0, c_head = a.fresh ? nhh + 1 : nhh;        // first chunk of the hidden layers / of the head blocks
  if (sec == 0) {
    const int kr = 16 * a.ktq;
    const float *loc = a.theta + a.woff[0], *bias = loc + 2 * q * 64, *dw = dwc + a.eoff[0];
    float *d1 = a.fresh ? chunks : a.blob + m.d1;
    for (int i = gtid; i < 4 * kr * 16; i += gsz) {
      const int jj = i & 15, sl = (i >> 4) % kr, t = (i >> 4) / kr;
      const int tt = sl >> 4, gg = (sl >> 2) & 3, rr = sl & 3, f = 16 * tt + 4 * rr + gg;     // slot 16t + 4g + r holds input feature 16t + 4r + g
      if (f < q) {
        if (slot == 0) a.blob[m.w1 + (t * kr + sl) * 17 + jj] = loc[f * 64 + 16 * t + jj];
        d1[(t * kr + sl) * 17 + jj] = dw[f * 64 + 16 * t + jj];
      }
    }
    if (slot == 0) {
      for (int i = gtid; i < 64; i += gsz) a.blob[m.b1 + i] = bias[i];
      for (int c = gtid; c < q; c += gsz) {
        const float scale = a.bnp[c] / sqrtf(a.bnp[3 * q + c] + 1e-3f);
        a.blob[m.sc + c] = scale; a.blob[m.sh + c] = a.bnp[q + c] - a.bnp[2 * q + c] * scale;
      }
    }
  } else if (sec < nh) {
    const int l = sec;
    const float *loc = a.theta + a.woff[l], *bias = loc + 2 * 4096, *dw = dwc + a.eoff[l];
    float *wl = a.blob + m.wh + (l - 1) * (4 * 64 * 17), *wd = chunks + (long long)(c_hid + l - 1) * BGMF_CHUNK;
    for (int i = gtid; i < 4 * 64 * 16; i += gsz) {
      const int jj = i & 15, rho = (i >> 4) & 63, t = i >> 10;
      if (slot == 0) wl[(t * 64 + rho) * 17 + jj] = loc[rho * 64 + 16 * t + jj];
      wd[(t * 64 + rho) * 17 + jj] = dw[rho * 64 + 16 * t + jj];
    }
    if (slot == 0) for (int i = gtid; i < 64; i += gsz) a.blob[m.bh + (l - 1) * 64 + i] = bias[i];
  } else {
    for (int head = 0; head < 2; ++head) {
      const float *loc = a.theta + a.woff[nh + head], *bias = loc + 2 * 64 * p, *dw = dwc + a.eoff[nh + head];
      for (int i = gtid; i < m.ntx * 64 * 16; i += gsz) {
        const int jj = i & 15, rho = (i >> 4) & 63, tx = i >> 10, o = 16 * tx + jj;
        if (o < p) {
          float *c = chunks + (long long)(c_head + tx) * BGMF_CHUNK;
          c[head * 64 * 17 + rho * 17 + jj] = loc[rho * p + o];
          c[BGM_PAIR + head * 64 * 17 + rho * 17 + jj] = dw[rho * p + o];
        }
      }
      if (slot == 0) for (int o = gtid; o < p; o += gsz) a.blob[m.bhd + head * 16 * m.ntx + o] = bias[o];
    }
  }
}
