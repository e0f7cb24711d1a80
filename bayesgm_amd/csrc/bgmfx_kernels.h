// bgmfx_kernels.h -- split precision (f16x3) form of the frozen-noise HMC of BGM with the Bayesian generator (bgmf_kernels.h).
//
// replaces: the same path as bgmf_hmc_kernel<FRESH = false> (BGM.tfp_mcmc_sampler bgm/base.py:709-830 on the target bgm/base.py:665-705
// with g_net = BayesianVariationalNet, networks/bnn.py:40-99; oracle/bgm_bnn.py hmc_sampler(frozen=True)), opt-in
// (bgm_bvn_set_precision(h, 2); params['hmc_precision'] = 'f16x3').  The arithmetic is the deterministic generator's split precision
// (bgm_kernels.h "PREC 2"): every operand x = x_hi + x_lo in fp16, W h ~ W_lo h_hi + W_hi h_lo + W_hi h_hi on the fp16 matrix
// instruction (K = 32), fp32 accumulation; likelihood, leapfrog and sign arithmetic stay fp32.  A Flipout layer
//     y = h loc + ((h * s_in) dW) * s_out + b
// is two such products per direction.  Nothing of the generator is LDS-resident: posterior means AND the run's perturbation are
// streamed as packed fp16 fragments, one step = [16 KiB unit of loc | 16 KiB unit of dW] for one layer and direction (or one
// 16-feature block of both heads, forward and transposed fragments), 2 nh - 1 + ntx steps per gradient evaluation through the
// double-buffered stage of BgmHeadStreamX3 (filled by global_load_lds, one barrier per step):
//     L1 (forward + transposed) | hidden 1 .. nh-1 forward | head blocks 0 .. ntx-1 | hidden nh-1 .. 1 transposed
// (L1 stays current from the last backward layer of one evaluation to the first forward layer of the next).
// The per-row signs multiply OPERANDS: s_in flips the packed fp16 halves of the split input (one XOR per two values; the split of
// -x is -x_hi, -x_lo exactly), s_out flips the fp32 result.  Transposed head products: the mean and variance parts of dW^T d must stay
// apart until the loop's end (their input signs differ), so each runs over a K block whose other half is zero.
// Measured (N = 2e5, p = 500, 10 leapfrog steps, one MI355X): 13.9 ms per transition (bgmf_hmc_kernel, fp32) -> 5.9 ms; DESIGN.md section 4h.
#pragma once
#include "bgmf_kernels.h"

static_assert(BGM_X3_STEP == 2, "a step of the Bayesian stream = [loc unit | dW unit]");
#ifndef BGMFX_WAVES
#define BGMFX_WAVES 8
#endif
// where in a head block's step the next step's direct-to-LDS loads are issued: 0 top, 1 behind the forward products, 2 before the
// barrier, 3 / 4 around the transposed loc products; ms per transition at C4's shape: 6.02 / 5.89 / (latency exposed) / ~5.9
#ifndef BGMFX_FETCH_AT
#define BGMFX_FETCH_AT 1
#endif

// XOR masks of the packed halves: element u of K block b <-> bit 8 b + u of `mask` (= bit 4t + r of a register-tile mask, t = 2b + (u >> 2), r = u & 3)
__device__ __forceinline__ bgm_u4 bgmfx_xmask(unsigned mask, int b) {
  bgm_u4 x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned bits = mask >> (8 * b + 2 * k);
    x[k] = ((bits & 1u) << 15) | ((bits & 2u) << 30);
  }
  return x;
}
__device__ __forceinline__ bgm_h8 bgmfx_xor(const bgm_h8 &v, const bgm_u4 &x) {
  return __builtin_bit_cast(bgm_h8, __builtin_bit_cast(bgm_u4, v) ^ x);
}
// (xh, xl) * s for the two K blocks of a 64-wide operand
__device__ __forceinline__ void bgmfx_flip2(const bgm_h8 (&xh)[2], const bgm_h8 (&xl)[2], unsigned mask, bgm_h8 (&fh)[2], bgm_h8 (&fl)[2]) {
  mask = bgmf_here(mask);
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const bgm_u4 x = bgmfx_xmask(mask, b);
    fh[b] = bgmfx_xor(xh[b], x);
    fl[b] = bgmfx_xor(xl[b], x);
  }
}
// acc[mt] += W x for one 64 x 64 unit (forward or transposed fragments, f = 2 (2 mt + b) (+1: lo))
__device__ __forceinline__ void bgmfx_layer(const unsigned char *unit, int lane, const bgm_h8 (&xh)[2], const bgm_h8 (&xl)[2], f32x4 (&acc)[4]) {
  const bgm_h8 *fr = reinterpret_cast<const bgm_h8 *>(unit) + lane;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    bgm_h8 ah[4], al[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { ah[mt] = fr[64 * (2 * (2 * mt + b))]; al[mt] = fr[64 * (2 * (2 * mt + b) + 1)]; }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = BGM_MFMA_H(al[mt], xh[b], acc[mt]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = BGM_MFMA_H(ah[mt], xl[b], acc[mt]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = BGM_MFMA_H(ah[mt], xh[b], acc[mt]);
  }
}
// the four forward chains of a head block (f = 2 (2 head + b) (+1)): mean head reads (mh, ml), variance head (vh, vl)
__device__ __forceinline__ void bgmfx_heads_fwd(const unsigned char *unit, int lane, const bgm_h8 (&mh)[2], const bgm_h8 (&ml)[2],
                                                const bgm_h8 (&vh)[2], const bgm_h8 (&vl)[2], f32x4 (&part)[4]) {
  const bgm_h8 *fr = reinterpret_cast<const bgm_h8 *>(unit) + lane;
  bgm_h8 ah[4], al[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) { ah[f] = fr[64 * (2 * f)]; al[f] = fr[64 * (2 * f + 1)]; }
#pragma unroll
  for (int f = 0; f < 4; ++f) part[f] = BGM_MFMA_H(al[f], f < 2 ? mh[f & 1] : vh[f & 1], part[f]);
#pragma unroll
  for (int f = 0; f < 4; ++f) part[f] = BGM_MFMA_H(ah[f], f < 2 ? ml[f & 1] : vl[f & 1], part[f]);
#pragma unroll
  for (int f = 0; f < 4; ++f) part[f] = BGM_MFMA_H(ah[f], f < 2 ? mh[f & 1] : vh[f & 1], part[f]);
}

// log p(z | x_obs) and dlogp/dz of the wave's 16 chains (z_dim <= 16: one latent tile).  On entry step 0 of the stream (L1) is current;
// every wave of the workgroup calls this the same number of times.  want_lp = false: the gradient alone (inner leapfrog steps).
template <int NH, int WAVES, bool X4>
__device__ __forceinline__ void bgmfx_logp_grad(const float *lds, const BgmfMeta &m, int j, int g, const f32x4 &z, const float *xrow,
                                                BgmHeadStreamX3<WAVES, X4> &hs, const unsigned *sg_row, float &logp, f32x4 &grad, bool want_lp) {
  const int lane = 16 * g + j;
  const f32x4 zero4 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  // The lane's sign masks are gathered from the wave's sign rows in LDS where they are used (two words and ~10 VALU operations per
  // mask): held in registers across the head loop the twelve of them were spilled.
  auto in_mask = [&](int l) { return bgmf_mask64(sg_row, m.sin_w[l], g); };
  auto out_mask = [&](int l) { return bgmf_mask64(sg_row, m.sout_w[l], g); };
  unsigned sgn[NH];
  f32x4 h[4];
  hs.fetch(1);
  {
    f32x4 zin;
#pragma unroll
    for (int r = 0; r < 4; ++r) zin[r] = fmaf(z[r], lds[m.sc + 4 * r + g], lds[m.sh + 4 * r + g]);      // input BatchNorm (inference mode)
    bgm_h8 zh, zl;
    bgm_split8(zin, zero4, zh, zl);      // k-slot u < 4 <-> latent feature 4 u + g
    unsigned m0 = 0u;      // layer 0 reads z: latent feature 4 r + g in register r (bgmf_signs)
    {
      const unsigned w = sg_row[m.sin_w[0]];
#pragma unroll
      for (int r = 0; r < 4; ++r) m0 |= ((w >> (4 * r + g)) & 1u) << r;
    }
    const bgm_u4 x = bgmfx_xmask(m0, 0);
    const bgm_h8 fh = bgmfx_xor(zh, x), fl = bgmfx_xor(zl, x);
    const bgm_h8 *frl = reinterpret_cast<const bgm_h8 *>(hs.tile(0)) + lane, *frd = reinterpret_cast<const bgm_h8 *>(hs.tile(1)) + lane;
    f32x4 al[4], ad[4];
    bgm_h8 ah[4], aw[4], dh_[4], dw_[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      ah[mt] = frl[64 * (2 * mt)]; aw[mt] = frl[64 * (2 * mt + 1)];
      dh_[mt] = frd[64 * (2 * mt)]; dw_[mt] = frd[64 * (2 * mt + 1)];
      al[mt] = *reinterpret_cast<const f32x4 *>(lds + m.b1 + 16 * mt + 4 * g);
      ad[mt] = zero4;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { al[mt] = BGM_MFMA_H(aw[mt], zh, al[mt]); ad[mt] = BGM_MFMA_H(dw_[mt], fh, ad[mt]); }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { al[mt] = BGM_MFMA_H(ah[mt], zl, al[mt]); ad[mt] = BGM_MFMA_H(dh_[mt], fl, ad[mt]); }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { al[mt] = BGM_MFMA_H(ah[mt], zh, al[mt]); ad[mt] = BGM_MFMA_H(dh_[mt], fh, ad[mt]); }
    bgmf_activate(al, ad, out_mask(0), sgn[0], h);
  }
  hs.commit();
#pragma unroll
  for (int l = 1; l < NH; ++l) {
    BGM_NO_HOIST();
    hs.fetch(l + 1);
    bgm_h8 xh[2], xl[2], fh[2], fl[2];
    bgm_split8(h[0], h[1], xh[0], xl[0]);
    bgm_split8(h[2], h[3], xh[1], xl[1]);
    bgmfx_flip2(xh, xl, in_mask(l), fh, fl);
    f32x4 al[4], ad[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { al[mt] = *reinterpret_cast<const f32x4 *>(lds + m.bh + (l - 1) * 64 + 16 * mt + 4 * g); ad[mt] = zero4; }
    bgmfx_layer(hs.tile(0), lane, xh, xl, al);
    bgmfx_layer(hs.tile(1), lane, fh, fl, ad);
    bgmf_activate(al, ad, out_mask(l), sgn[l], h);
    hs.commit();
  }
  // heads, one 16-feature block of both (loc and dW) per step
  float nll = 0.0f;
  f32x4 dh[4], dhm[4], dhv[4];
  bgmf_zero<4>(dh); bgmf_zero<4>(dhm); bgmf_zero<4>(dhv);
  {
    bgm_h8 hh[2], hl[2];
    bgm_split8(h[0], h[1], hh[0], hl[0]);
    bgm_split8(h[2], h[3], hh[1], hl[1]);
    bgm_u4 xm_[2], xv_[2];      // XOR masks of the heads' input signs (per evaluation); the flipped operands are made per block
    {
      const unsigned im = in_mask(NH), iv = in_mask(NH + 1);
#pragma unroll
      for (int b = 0; b < 2; ++b) { xm_[b] = bgmfx_xmask(im, b); xv_[b] = bgmfx_xmask(iv, b); }
    }
#pragma unroll 1
    for (int tx = 0; tx < m.ntx; ++tx) {
      BGM_NO_HOIST();
      if (!hs.x_valid) { hs.load_x1(xrow, m.p, 0, g); hs.x_valid = true; }      // (the first evaluation of a tile only)
      const f32x4 xv = hs.take_x(0);
      const int next_step = tx + 1 < m.ntx ? NH + tx + 1 : (NH > 1 ? NH + m.ntx : 0);
#if BGMFX_FETCH_AT == 0
      hs.fetch(next_step);
#endif
      hs.load_x1(xrow, m.p, tx + 1 < m.ntx ? tx + 1 : 0, g);                     // the next block's data values, a step ahead
      asm volatile("" ::: "memory");
      f32x4 part[4], pd[4];
      part[0] = *reinterpret_cast<const f32x4 *>(lds + m.bhd + 16 * tx + 4 * g);
      part[2] = *reinterpret_cast<const f32x4 *>(lds + m.bhd + 16 * (m.ntx + tx) + 4 * g);
      part[1] = zero4; part[3] = zero4;
      bgmf_zero<4>(pd);
      bgmfx_heads_fwd(hs.tile(0), lane, hh, hl, hh, hl, part);
      BGM_NO_HOIST();
      {
        bgm_h8 mh[2], ml[2], vh[2], vl[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          mh[b] = bgmfx_xor(hh[b], xm_[b]); ml[b] = bgmfx_xor(hl[b], xm_[b]);
          vh[b] = bgmfx_xor(hh[b], xv_[b]); vl[b] = bgmfx_xor(hl[b], xv_[b]);
        }
        bgmfx_heads_fwd(hs.tile(1), lane, mh, ml, vh, vl, pd);
      }
      BGM_NO_HOIST();
#if BGMFX_FETCH_AT == 1
      hs.fetch(next_step);
#endif
      const int sh_ = 16 * (tx & 1) + 4 * g;
      const unsigned bm = (sg_row[m.sout_w[NH] + (tx >> 1)] >> sh_) & 0xFu, bv = (sg_row[m.sout_w[NH + 1] + (tx >> 1)] >> sh_) & 0xFu;
      f32x4 dms[2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d0 = pd[0][r] + pd[1][r], d1 = pd[2][r] + pd[3][r];
        const float mu = part[0][r] + part[1][r] + bgmf_flip(d0, (bm >> r) & 1u), sr = part[2][r] + part[3][r] + bgmf_flip(d1, (bv >> r) & 1u);
        float dmu, ds;
        bgm_x3_lik<true>(xv[r], mu, sr, 16 * tx + 4 * g + r < m.p, want_lp, nll, dmu, ds);
        dms[0][r] = dmu; dms[1][r] = ds;
      }
      bgm_h8 dhi, dlo;
      bgm_split8(dms[0], dms[1], dhi, dlo);
#if BGMFX_FETCH_AT == 3
      hs.fetch(next_step);
#endif
      bgm_x3_bwd(hs.tile(0), lane, dhi, dlo, dh);
      BGM_NO_HOIST();
#if BGMFX_FETCH_AT == 4
      hs.fetch(next_step);
#endif
      {
        // d * s_out in packed halves; the mean part in the lower half of one K block, the variance part in the upper half of another
        const bgm_u4 hi = __builtin_bit_cast(bgm_u4, dhi), lo = __builtin_bit_cast(bgm_u4, dlo);
        const unsigned x0 = ((bm & 1u) << 15) | ((bm & 2u) << 30), x1 = ((bm & 4u) << 13) | ((bm & 8u) << 28);
        const unsigned x2 = ((bv & 1u) << 15) | ((bv & 2u) << 30), x3 = ((bv & 4u) << 13) | ((bv & 8u) << 28);
        const bgm_h8 mhi = __builtin_bit_cast(bgm_h8, bgm_u4{hi[0] ^ x0, hi[1] ^ x1, 0u, 0u}), mlo = __builtin_bit_cast(bgm_h8, bgm_u4{lo[0] ^ x0, lo[1] ^ x1, 0u, 0u});
        const bgm_h8 vhi = __builtin_bit_cast(bgm_h8, bgm_u4{0u, 0u, hi[2] ^ x2, hi[3] ^ x3}), vlo = __builtin_bit_cast(bgm_h8, bgm_u4{0u, 0u, lo[2] ^ x2, lo[3] ^ x3});
        const bgm_h8 *fr = reinterpret_cast<const bgm_h8 *>(hs.tile(1)) + lane;
        bgm_h8 ah[4], al[4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) { ah[ti] = fr[64 * (8 + 2 * ti)]; al[ti] = fr[64 * (9 + 2 * ti)]; }
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) { dhm[ti] = BGM_MFMA_H(al[ti], mhi, dhm[ti]); dhv[ti] = BGM_MFMA_H(al[ti], vhi, dhv[ti]); }
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) { dhm[ti] = BGM_MFMA_H(ah[ti], mlo, dhm[ti]); dhv[ti] = BGM_MFMA_H(ah[ti], vlo, dhv[ti]); }
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) { dhm[ti] = BGM_MFMA_H(ah[ti], mhi, dhm[ti]); dhv[ti] = BGM_MFMA_H(ah[ti], vhi, dhv[ti]); }
      }
#if BGMFX_FETCH_AT == 2
      hs.fetch(next_step);
#endif
      hs.commit();
    }
  }
  float zsq = 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) zsq = fmaf(z[r], z[r], zsq);
  logp = -sum_over_g(nll + 0.5f * zsq);
  {
    const unsigned im = in_mask(NH), iv = in_mask(NH + 1);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = dhm[t][r], b = dhv[t][r];
        dh[t][r] += bgmf_flip(a, (im >> (4 * t + r)) & 1u) + bgmf_flip(b, (iv >> (4 * t + r)) & 1u);
      }
  }
#pragma unroll
  for (int l = NH - 1; l >= 0; --l) {
    BGM_NO_HOIST();
    if (l > 0) hs.fetch(l > 1 ? NH + m.ntx + (NH - 1 - l) + 1 : 0);
    // d(pre-activation) = dh o LeakyReLU', clamped into the fp16 range, split once; its s_out-flipped copy for the dW product
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        dh[t][r] = __builtin_amdgcn_fmed3f(dh[t][r] * (((sgn[l] >> (4 * t + r)) & 1u) ? 1.0f : BGM_LEAK), -6.0e4f, 6.0e4f);
    bgm_h8 ph[2], pl[2], sh2[2], sl2[2];
    bgm_split8(dh[0], dh[1], ph[0], pl[0]);
    bgm_split8(dh[2], dh[3], ph[1], pl[1]);
    bgmfx_flip2(ph, pl, out_mask(l), sh2, sl2);
    if (l > 0) {
      f32x4 dn[4], dd[4];
      bgmf_zero<4>(dn); bgmf_zero<4>(dd);
      bgmfx_layer(hs.tile(0), lane, ph, pl, dn);
      bgmfx_layer(hs.tile(1), lane, sh2, sl2, dd);
      const unsigned il = in_mask(l);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = dd[t][r];
          dh[t][r] = dn[t][r] + bgmf_flip(d, (il >> (4 * t + r)) & 1u);
        }
      hs.commit();
    } else {      // first layer (step 0 is current again): the latent gradient, rows of the transposed fragments in the latent layout
      const bgm_h8 *frl = reinterpret_cast<const bgm_h8 *>(hs.tile(0)) + lane, *frd = reinterpret_cast<const bgm_h8 *>(hs.tile(1)) + lane;
      f32x4 ga[2], gd[2];
      bgm_h8 ah[2], aw[2], dh_[2], dw_[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        ah[b] = frl[64 * (8 + 2 * b)]; aw[b] = frl[64 * (9 + 2 * b)];
        dh_[b] = frd[64 * (8 + 2 * b)]; dw_[b] = frd[64 * (9 + 2 * b)];
        ga[b] = zero4; gd[b] = zero4;
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) { ga[b] = BGM_MFMA_H(aw[b], ph[b], ga[b]); gd[b] = BGM_MFMA_H(dw_[b], sh2[b], gd[b]); }
#pragma unroll
      for (int b = 0; b < 2; ++b) { ga[b] = BGM_MFMA_H(ah[b], pl[b], ga[b]); gd[b] = BGM_MFMA_H(dh_[b], sl2[b], gd[b]); }
#pragma unroll
      for (int b = 0; b < 2; ++b) { ga[b] = BGM_MFMA_H(ah[b], ph[b], ga[b]); gd[b] = BGM_MFMA_H(dh_[b], sh2[b], gd[b]); }
      const unsigned w0 = sg_row[m.sin_w[0]];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = gd[0][r] + gd[1][r];
        const float gz = ga[0][r] + ga[1][r] + bgmf_flip(d, (w0 >> (4 * r + g)) & 1u);
        grad[r] = fmaf(gz, lds[m.sc + 4 * r + g], -z[r]);   // through the input affine; prior -|z|^2 / 2
      }
    }
  }
}

struct BgmfxHmcKArgs {
  BgmfHmcKArgs k;              // k.blob: the fp32 resident part [b1 | bh | bhd | sc | sh] in k.m's layout; k.m.stage / sign: LDS offsets
  const unsigned char *sx;     // the stream: [2 nh - 1 + ntx][2][BGM_X3_BLOCK_BYTES]
};

// The transition logic of bgmf_hmc_kernel<1, NH, false> on the streamed split-precision target.  X4: x_dim % 4 == 0 (one 16-byte request
// per lane and block for the data values; else four clamped 4-byte requests).
template <int NH, int WAVES, bool X4>
__global__ __launch_bounds__(64 * WAVES) void bgmfx_hmc_kernel(BgmfxHmcKArgs xa) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BgmfHmcKArgs &a = xa.k;
  const BgmfMeta &m = a.m;
  lds_fill(lds, a.blob, m.resident);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int evals = (a.init ? 1 : 0) + a.n_iters * a.n_leapfrog;
  BgmHeadStreamX3<WAVES, X4> hs;
  hs.begin_at(xa.sx, lds + m.stage);
  unsigned *sg_row = reinterpret_cast<unsigned *>(lds + m.sign) + (16 * wave + j) * m.swp;
  const long long n = a.n, n_tiles = (n + 15) / 16, passes = bgm_block_passes(n_tiles, WAVES);
  const float eps = *a.step;
  const int n_steps = 2 * NH - 1 + m.ntx;
  for (long long ps = 0; ps < passes; ++ps) {
    long long tile = (ps * WAVES + wave) * gridDim.x + blockIdx.x;      // wave-major, as bgmf_hmc_kernel
    const bool tile_ok = tile < n_tiles;
    tile = tile_ok ? tile : n_tiles - 1;
    long long row = tile * 16 + j;
    const bool ok = tile_ok && row < n;
    row = row < n ? row : n - 1;
    const unsigned rowid = (unsigned)(a.row_base + row);
    const float *xrow = a.x + row * (long long)m.p;
    __syncthreads();         // (the previous pass's readers of the sign rows are done)
    for (int c = g; c < (m.swords >> 2); c += 4) {      // the wave's 16 sign rows (bgmf_signs, noise stream 0)
      const uint4 w4 = philox4x32_10(rowid, (unsigned)c, 0u, BNN_TAG_SIGN, a.k0, a.k1);
      sg_row[4 * c] = w4.x; sg_row[4 * c + 1] = w4.y; sg_row[4 * c + 2] = w4.z; sg_row[4 * c + 3] = w4.w;
    }
    __syncthreads();
    if (!tile_ok) {          // no tile: keep the workgroup's stream moving
      for (int e = 0; e < evals; ++e)
        for (int k = 0; k < n_steps; ++k) { hs.fetch(k + 1 < n_steps ? k + 1 : 0); hs.commit(); }
      continue;
    }
    hs.x_valid = false;      // (a new row: nothing of it has been requested ahead)
    f32x4 z, gr;
    float lp;
    if (a.init) {   // initial_state ~ N(0,1)  (bgm/base.py:778), RNG tag 0
      const f32x4 e = box_muller4(philox4x32_10(rowid, 0u, (unsigned)g, TAG_INIT, a.k0, a.k1));
#pragma unroll
      for (int r = 0; r < 4; ++r) z[r] = (4 * r + g < m.q) ? e[r] : 0.0f;
      bgmfx_logp_grad<NH, WAVES, X4>(lds, m, j, g, z, xrow, hs, sg_row, lp, gr, true);
    } else {
      f32x4 z1[1], g1[1];
      bgm_load_z<1>(a.state, m.q, row, g, z1);
      bgm_load_z<1>(a.grad, m.q, row, g, g1);
      z = z1[0]; gr = g1[0];
      lp = a.logp[row];
    }
    for (int it = a.it_begin; it < a.it_begin + a.n_iters; ++it) {
      BGM_NO_HOIST();
      f32x4 mom, zc, gc;
      float ke0 = 0.0f;
      {
        const f32x4 e = box_muller4(philox4x32_10(rowid, (unsigned)it, (unsigned)g, TAG_MOM, a.k0, a.k1));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pm = (4 * r + g < m.q) ? e[r] : 0.0f;
          ke0 = fmaf(pm, pm, ke0);
          mom[r] = fmaf(0.5f * eps, gr[r], pm);
          zc[r] = z[r];
        }
      }
      ke0 = sum_over_g(ke0);
      float lpc = lp;
      for (int l = 0; l < a.n_leapfrog; ++l) {
        BGM_NO_HOIST();
#pragma unroll
        for (int r = 0; r < 4; ++r) zc[r] = fmaf(eps, mom[r], zc[r]);
        bgmfx_logp_grad<NH, WAVES, X4>(lds, m, j, g, zc, xrow, hs, sg_row, lpc, gc, l == a.n_leapfrog - 1);
        const float kick = (l < a.n_leapfrog - 1) ? eps : 0.5f * eps;
#pragma unroll
        for (int r = 0; r < 4; ++r) mom[r] = fmaf(kick, gc[r], mom[r]);
      }
      float ke1 = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) ke1 = fmaf(mom[r], mom[r], ke1);
      ke1 = sum_over_g(ke1);
      float log_ratio = -((-lpc + 0.5f * ke1) - (-lp + 0.5f * ke0));
      log_ratio = (log_ratio == log_ratio && fabsf(log_ratio) != INFINITY) ? log_ratio : -INFINITY;
      const uint4 w4 = philox4x32_10(rowid, (unsigned)it >> 2, 0u, TAG_HACC, a.k0, a.k1);
      const unsigned w_ = (it & 2) ? ((it & 1) ? w4.w : w4.z) : ((it & 1) ? w4.y : w4.x);
      const float u = u01_open(w_);
      const bool acc = logf(u) < log_ratio;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        z[r] = acc ? zc[r] : z[r];
        gr[r] = acc ? gc[r] : gr[r];
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
      if (a.draws != nullptr && it >= a.burn_in && ok) {
        const f32x4 z1[1] = {z};
        bgm_store_z<1>(a.draws + (long long)(it - a.burn_in) * n * m.q, m.q, row, g, z1);
      }
    }
    if (ok) {
      const f32x4 z1[1] = {z}, g1[1] = {gr};
      bgm_store_z<1>(a.state, m.q, row, g, z1);
      bgm_store_z<1>(a.grad, m.q, row, g, g1);
      if (g == 0) a.logp[row] = lp;
    }
  }
}

// The stream and the resident part from the session's parameter vector and the run's perturbation (sources as bgmf_pack_kernel).
// grid (2 nh - 1 + ntx steps + 1, 16): y = 8 unit + fragment pair (hi at fragment 2 fp, lo behind it); 512 threads = 64 lanes x 8 slots.
// The extra x block writes the resident floats.  Fragment layouts: bgm_api.hip "the stream" (hidden, heads), L1 without the folded
// BatchNorm (the affine is applied to z in registers: the input signs multiply the NORMALISED input).
struct BgmfxPackArgs {
  BgmfMeta m;
  int woff[BGMF_MAXNH + 2], eoff[BGMF_MAXNH + 2];
  const float *bnp, *theta, *dwc;
  unsigned char *sx;
  float *res;
};
static __global__ __launch_bounds__(512) void bgmfx_pack_kernel(BgmfxPackArgs a) {
  const BgmfMeta &m = a.m;
  const int q = m.q, p = m.p, nh = m.nh, ntx = m.ntx, n_steps = 2 * nh - 1 + ntx;
  const int step = blockIdx.x;
  if (step == n_steps) {
    const int tid = blockIdx.y * blockDim.x + threadIdx.x, tsz = gridDim.y * blockDim.x;
    for (int i = tid; i < 64; i += tsz) a.res[m.b1 + i] = (a.theta + a.woff[0] + 2 * q * 64)[i];
    for (int i = tid; i < (nh - 1) * 64; i += tsz) a.res[m.bh + i] = (a.theta + a.woff[1 + i / 64] + 2 * 4096)[i & 63];
    for (int head = 0; head < 2; ++head)
      for (int o = tid; o < p; o += tsz) a.res[m.bhd + head * 16 * ntx + o] = (a.theta + a.woff[nh + head] + 2 * 64 * p)[o];
    for (int c = tid; c < q; c += tsz) {
      const float scale = a.bnp[c] / sqrtf(a.bnp[3 * q + c] + 1e-3f);
      a.res[m.sc + c] = scale; a.res[m.sh + c] = a.bnp[q + c] - a.bnp[2 * q + c] * scale;
    }
    return;
  }
  const int unit = blockIdx.y >> 3, fp = blockIdx.y & 7, lane = threadIdx.x >> 3, u = threadIdx.x & 7;
  const int i = lane & 15, gA = lane >> 4, ku = 4 * gA + (u & 3);
  auto src = [&](int layer) { return unit ? a.dwc + a.eoff[layer] : a.theta + a.woff[layer]; };
  float w = 0.0f;
  if (step == 0) {
    const float *W = src(0);
    if (fp < 4) {
      const int f = 4 * u + gA;
      w = (u < 4 && f < q) ? W[f * 64 + 16 * fp + i] : 0.0f;
    } else if (fp < 6) {
      const int f = 4 * (i & 3) + (i >> 2);
      w = f < q ? W[f * 64 + 16 * (2 * (fp - 4) + (u >> 2)) + ku] : 0.0f;
    }
  } else if (step < nh || step >= nh + ntx) {
    const bool fwd = step < nh;
    const int l = fwd ? step : nh - 1 - (step - nh - ntx);
    const float *W = src(l);
    const int t = fp >> 1, k = 16 * (2 * (fp & 1) + (u >> 2)) + ku;
    w = fwd ? W[k * 64 + 16 * t + i] : W[(16 * t + i) * 64 + k];
  } else {
    const int tx = step - nh;
    const float *Wm = src(nh), *Wv = src(nh + 1);
    if (fp < 4) {
      const int un = 16 * (2 * (fp & 1) + (u >> 2)) + ku, col = 16 * tx + i;
      w = col < p ? ((fp >> 1) ? Wv : Wm)[(long long)un * p + col] : 0.0f;
    } else {
      const int un = 16 * (fp - 4) + i, col = 16 * tx + 4 * gA + (u & 3);
      w = col < p ? (u < 4 ? Wm : Wv)[(long long)un * p + col] : 0.0f;
    }
  }
  w = fminf(fmaxf(w, -65504.0f), 65504.0f);
  const _Float16 hi = (_Float16)w, lo = (_Float16)(w - (float)hi);
  _Float16 *dst = reinterpret_cast<_Float16 *>(a.sx + ((long long)step * 2 + unit) * BGM_X3_BLOCK_BYTES) + (2 * fp) * 512 + lane * 8 + u;
  dst[0] = hi; dst[512] = lo;
}
