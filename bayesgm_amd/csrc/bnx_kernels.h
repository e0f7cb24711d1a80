// bnx_kernels.h -- split-precision ("f16 x 3") form of the Bayesian-network sampling path of bnf_kernels.h (gfx950), opt-in
// (`bgm_bnn_set_precision(h, 2)`, params['mh_precision'] = 'f16x3' on a use_bnn=True model); fp32 (bnf_kernels.h) stays the default.
//
// replaces (src/bayesgm/models/causalbgm/base.py, use_bnn branches; networks/bnn.py:4-38): the same functions as bnf_kernels.h --
// get_log_posterior :765-817, metropolis_hastings_sampler :820-904, infer_from_latent_posterior :671-763 -- with the same kernels
// around the networks (bnf_mh_kernel<..., X3 = true>, bnf_effects_kernel<..., X3 = true>: item queue, proposal, accept rule, Philox
// streams, sign words and perturbation draws are bnf_kernels.h's own code); only the Flipout layers change:
//     y = loc^T h + ((dW^T (h o s_in)) o s_out) + b
// with every operand a sum of two fp16 numbers (x = x_hi + x_lo, 22 mantissa bits) and each of the two contractions
//     W h ~= W_lo h_hi + W_hi h_lo + W_hi h_hi                    (fp32 accumulation; the W_lo h_lo term ~2^-22 |W h| is dropped)
// on v_mfma_f32_16x16x32_f16: a 64 -> 64 Flipout layer is 48 matrix instructions of 16 cycles instead of 128 of 32.
//
// Layout.  M = output feature, N = row (16 per tile), K = input feature as in bnf_kernels.h, so a layer's accumulators (lane (j, g),
// tile t, register r = feature 16 t + 4 g + r of row j) are the next layer's B operands once packed: a K block of 32 inputs is the pair
// of tiles (2T, 2T + 1); lane group g supplies k-slots 8 g + u with u = 2 r + s  <->  feature 16 (2T + s) + 4 g + r.  That pairing puts
// the two features of one packed register 16 bits apart in the layer's 32-feature Rademacher word (bits 4 g + r and 16 + 4 g + r), so
//   * the input flip h o s_in is ONE mask (word << k) & 0x80008000 per register pair and an XOR on the hi and on the lo word (a sign
//     flip is exact in both), and
//   * the output flip is one v_and_or that turns the same two bits into a packed (+-1, +-1) fp16 pair and a v_fma_mix_f32 per element.
// A 16-input tail (the first layers at q + 1 <= 16, the 8 -> 2 output layers) uses v_mfma_f32_16x16x16_f16, k-slot 4 g + r.
// Memory.  A K = 32 block of one output tile is [hi: 64 lanes x 16 B | lo: 64 lanes x 16 B] = 2 KB = the two fp32 fragments it
// replaces; a K = 16 block is [hi: 64 x 8 B | lo: 64 x 8 B] = 1 KB = its one fp32 fragment: the blob, the perturbation sets, every
// fragment index of BnfPlan and the bias / normalisation / shift tables keep their offsets (bnx_api.hip re-encodes the fragments,
// bnx_noise_kernel writes dW = sigma * eps as hi / lo at the positions of the second table).
#pragma once
#include "bnf_kernels.h"

typedef _Float16 bnx_h;
typedef bnx_h bnx_h8 __attribute__((ext_vector_type(8)));
typedef bnx_h bnx_h4 __attribute__((ext_vector_type(4)));
typedef bnx_h bnx_h2 __attribute__((ext_vector_type(2)));
typedef float bnx_f2 __attribute__((ext_vector_type(2)));
typedef unsigned bnx_u4 __attribute__((ext_vector_type(4)));
typedef unsigned bnx_u2 __attribute__((ext_vector_type(2)));

#define BNX_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(bnx_h8, (a)), __builtin_bit_cast(bnx_h8, (b)), (c), 0, 0, 0)
#define BNX_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(bnx_h4, (a)), __builtin_bit_cast(bnx_h4, (b)), (c), 0, 0, 0)

// one K = 32 block of activations of one row tile: register r = features (16 (2T) + 4 g + r, 16 (2T + 1) + 4 g + r) as (hi, lo) fp16 pairs
struct BnxK { bnx_u4 hi, lo; };
// one K = 16 block: register pr = slots (2 pr, 2 pr + 1)
struct BnxK16 { bnx_u2 hi, lo; };

// constants of the sign arithmetic in SGPRs (VOP3 takes no literal on gfx9; see bnf_k80)
__device__ __forceinline__ uint32_t bnx_k8() { uint32_t k; asm("s_mov_b32 %0, 0x80008000" : "=s"(k)); return k; }
// (0x3c003c00 sits in a VGPR: a VOP3 instruction reads one SGPR, and (t & k8) | k3c is a single v_bitop3 / v_and_or only then)
__device__ __forceinline__ uint32_t bnx_k3c() { uint32_t k; asm("v_mov_b32 %0, 0x3c003c00" : "=v"(k)); return k; }

// (a, b) -> packed fp16 (hi pair, lo pair), a = hi_a + lo_a up to 2^-22 |a|: 3 VALU instructions per element
__device__ __forceinline__ void bnx_split_pair(float a, float b, unsigned &hi, unsigned &lo) {
  const bnx_h2 h = __builtin_convertvector(bnx_f2{a, b}, bnx_h2);
  hi = __builtin_bit_cast(unsigned, h);
  const float la = a - (float)h[0], lb = b - (float)h[1];
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(bnx_f2{la, lb}, bnx_h2));
}

// Epilogue of an output tile PAIR (2T, 2T + 1) of a Flipout layer behind a LeakyReLU: y = a1 + s_out a2, v = lrelu_s(y) as K block T of the
// next layer.  wo_sh: the pair's output-sign word, pre-shifted (bnf_preshift: bit 4 g + 3 at position 15, bit 16 + 4 g + 3 at position 31).
__device__ __forceinline__ void bnx_epi_pair(const f32x4 &a1e, const f32x4 &a2e, const f32x4 &a1o, const f32x4 &a2o, uint32_t wo_sh, BnxK &h) {
  const uint32_t k8 = bnx_k8(), k3c = bnx_k3c();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t to = r == 3 ? wo_sh : wo_sh << (3 - r);
    const bnx_h2 pm = __builtin_bit_cast(bnx_h2, (to & k8) | k3c);          // (+-1, +-1): even tile's sign in the low half
    const float ye = fmaf(a2e[r], (float)pm[0], a1e[r]), yo = fmaf(a2o[r], (float)pm[1], a1o[r]);
    unsigned hi, lo;
    bnx_split_pair(lrelu_s(ye), lrelu_s(yo), hi, lo);
    h.hi[r] = hi; h.lo[r] = lo;
  }
}
// hs = h o s_in for one K block; wi_sh: the block's input-sign word, pre-shifted.  Formed where the block is consumed (the consuming layer's
// start), not where it is produced: a layer in flight then holds h, hs of its input and only h of its output.
__device__ __forceinline__ void bnx_flip(const BnxK &h, uint32_t wi_sh, BnxK &hs) {
  const uint32_t k8 = bnx_k8();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t m = (r == 3 ? wi_sh : wi_sh << (3 - r)) & k8;
    hs.hi[r] = h.hi[r] ^ m; hs.lo[r] = h.lo[r] ^ m;
  }
}

// fragment reads.  LF / DW point at a layer's first fragment (units of f32x4 = 16 B; fragment f at + 64 f).  A K = 32 block (t even) of an
// output tile: hi at fragment t, lo at fragment t + 1 (one f32x4 per lane each); a K = 16 block: hi = 8 B per lane in the first half of
// its fragment, lo in the second half.
__device__ __forceinline__ bnx_u4 bnx_ld32(const f32x4 *F, int frag, int lane) { return __builtin_bit_cast(bnx_u4, F[frag * 64 + lane]); }
__device__ __forceinline__ bnx_u2 bnx_ld16(const f32x4 *F, int frag, int lane, int lo) {
  return reinterpret_cast<const bnx_u2 *>(F + frag * 64)[64 * lo + lane];
}

// the (normalised, packed) extended input of R row tiles: one K = 16 block (KS <= 4) or one K = 32 block (KS <= 8), plain and flipped
template <int KS>
struct BnxIn {
  static constexpr int T0 = (KS + 3) / 4;
  static_assert(T0 <= 2, "bnx: at most 32 extended inputs");
  bnx_u4 hi, lo, his, los;      // T0 == 1: components 0, 1 only
};

// normalise (scale / shift per slot and net), split and flip the extended input.  Slot ks = 4 sb + r of a lane is input 16 sb + 4 r + g.
// K = 16: register pr packs slots (2 pr, 2 pr + 1); K = 32: register r packs slots (r, 4 + r) (u = 2 r + s).
template <int KS, int R>
__device__ __forceinline__ void bnx_input(const f32x4 *NORM, const int4 *SHIFT, int g, const float (&ze)[R][KS], const uint32_t (&w_in)[R],
                                          BnxIn<KS> (&in)[R]) {
  constexpr int T0 = (KS + 3) / 4;
  float hb[R][4 * T0];
  uint32_t sb31[R][4 * T0];      // the slot's input sign at bit 31
#pragma unroll
  for (int sb = 0; sb < T0; ++sb) {
    const f32x4 sc = NORM[(sb * 2) * 4 + g], sh = NORM[(sb * 2 + 1) * 4 + g];
    const int4 st = SHIFT[sb * 4 + g];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ks = 4 * sb + r;
      const int sft = r == 0 ? st.x : r == 1 ? st.y : r == 2 ? st.z : st.w;
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        if (ks < KS) { hb[rt][ks] = fmaf(ze[rt][ks], sc[r], sh[r]); sb31[rt][ks] = (w_in[rt] << sft) & 0x80000000u; }
        else { hb[rt][ks] = 0.0f; sb31[rt][ks] = 0u; }
      }
    }
  }
#pragma unroll
  for (int rt = 0; rt < R; ++rt) {
    if constexpr (T0 == 1) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        unsigned hi, lo;
        bnx_split_pair(hb[rt][2 * pr], hb[rt][2 * pr + 1], hi, lo);
        const uint32_t m = (sb31[rt][2 * pr] >> 16) | sb31[rt][2 * pr + 1];
        in[rt].hi[pr] = hi; in[rt].lo[pr] = lo; in[rt].his[pr] = hi ^ m; in[rt].los[pr] = lo ^ m;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        unsigned hi, lo;
        bnx_split_pair(hb[rt][r], hb[rt][4 + r], hi, lo);
        const uint32_t m = (sb31[rt][r] >> 16) | sb31[rt][4 + r];
        in[rt].hi[r] = hi; in[rt].lo[r] = lo; in[rt].his[r] = hi ^ m; in[rt].los[r] = lo ^ m;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// first layer: packed extended input -> 64 units (bnf_first).  wo: output-sign words (pre-shifted).
// ---------------------------------------------------------------------------------------------
template <int KS, int R>
__device__ __forceinline__ void bnx_first(const f32x4 *LF, const f32x4 *__restrict__ DW, const f32x4 *BL, int lane, int g, const BnxIn<KS> (&in)[R],
                                          const uint32_t (&wo)[R][2], BnxK (&h)[R][2]) {
  constexpr int T0 = (KS + 3) / 4;
  bnx_u4 dh[4], dl[4];            // the four output tiles' perturbation fragments (K = 16: components 0, 1)
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    if constexpr (T0 == 1) {
      const bnx_u2 a = bnx_ld16(DW, mt, lane, 0), b = bnx_ld16(DW, mt, lane, 1);
      dh[mt] = bnx_u4{a[0], a[1], 0u, 0u}; dl[mt] = bnx_u4{b[0], b[1], 0u, 0u};
    } else { dh[mt] = bnx_ld32(DW, 2 * mt, lane); dl[mt] = bnx_ld32(DW, 2 * mt + 1, lane); }
  }
  BNF_PIN();
#pragma unroll
  for (int mp = 0; mp < 2; ++mp) {
    f32x4 a1[2][R], a2[2][R];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int mt = 2 * mp + e;
      const f32x4 b = BL[4 * mt + g];
#pragma unroll
      for (int rt = 0; rt < R; ++rt) { a1[e][rt] = b; a2[e][rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      if constexpr (T0 == 1) {
        const bnx_u2 ah = bnx_ld16(LF, mt, lane, 0), al = bnx_ld16(LF, mt, lane, 1);
        const bnx_u2 fh = bnx_u2{dh[mt][0], dh[mt][1]}, fl = bnx_u2{dl[mt][0], dl[mt][1]};
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
          const bnx_u2 bh = bnx_u2{in[rt].hi[0], in[rt].hi[1]}, bs = bnx_u2{in[rt].his[0], in[rt].his[1]};
          a1[e][rt] = BNX_MFMA16(al, bh, a1[e][rt]); a2[e][rt] = BNX_MFMA16(fl, bs, a2[e][rt]);
        }
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
          const bnx_u2 bl = bnx_u2{in[rt].lo[0], in[rt].lo[1]}, bs = bnx_u2{in[rt].los[0], in[rt].los[1]};
          a1[e][rt] = BNX_MFMA16(ah, bl, a1[e][rt]); a2[e][rt] = BNX_MFMA16(fh, bs, a2[e][rt]);
        }
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
          const bnx_u2 bh = bnx_u2{in[rt].hi[0], in[rt].hi[1]}, bs = bnx_u2{in[rt].his[0], in[rt].his[1]};
          a1[e][rt] = BNX_MFMA16(ah, bh, a1[e][rt]); a2[e][rt] = BNX_MFMA16(fh, bs, a2[e][rt]);
        }
      } else {
        const bnx_u4 ah = bnx_ld32(LF, 2 * mt, lane), al = bnx_ld32(LF, 2 * mt + 1, lane);
#pragma unroll
        for (int rt = 0; rt < R; ++rt) { a1[e][rt] = BNX_MFMA32(al, in[rt].hi, a1[e][rt]); a2[e][rt] = BNX_MFMA32(dl[mt], in[rt].his, a2[e][rt]); }
#pragma unroll
        for (int rt = 0; rt < R; ++rt) { a1[e][rt] = BNX_MFMA32(ah, in[rt].lo, a1[e][rt]); a2[e][rt] = BNX_MFMA32(dh[mt], in[rt].los, a2[e][rt]); }
#pragma unroll
        for (int rt = 0; rt < R; ++rt) { a1[e][rt] = BNX_MFMA32(ah, in[rt].hi, a1[e][rt]); a2[e][rt] = BNX_MFMA32(dh[mt], in[rt].his, a2[e][rt]); }
      }
    }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) bnx_epi_pair(a1[0][rt], a2[0][rt], a1[1][rt], a2[1][rt], wo[rt][mp], h[rt][mp]);
  }
}

// ---------------------------------------------------------------------------------------------
// The generator's hidden and last layers read their fragments as ONE linear stream (BnfPlan places g's last layer behind its hidden
// layers): output tile i of the stream (i = 4 (l - 1) + mt in hidden layer l, 16 + mt in the last layer) has its four perturbation
// fragments [K block 0 hi | lo | K block 1 hi | lo] at DWs + 4 i fragments and its posterior means at LFs + 4 i.  Perturbations come
// from L2: a tile's are requested TWO tiles ahead (one tile is 12 R matrix instructions, less than an L2 round trip) into a ring of
// three register sets; the means come from LDS one K block ahead.
// ---------------------------------------------------------------------------------------------
struct BnxRing { bnx_u4 cur[4], nxt[4]; bnx_u4 la[2]; };
// development ablations (never defined in the product build): BNX_ABL_NODW reads the stream's perturbation fragments from LDS (the means)
// instead of L2; BNX_ABL_NOV reads every data row from the block's first row (cache resident)
#ifdef BNX_ABL_NODW
#define BNX_DWSRC(dw, lf) (lf)
#else
#define BNX_DWSRC(dw, lf) (dw)
#endif
// (i: the stream's first tile behind the ring; n_tiles: its length, requests beyond it are clamped to the last tile)
__device__ __forceinline__ void bnx_ring_fill(BnxRing &rg, const f32x4 *__restrict__ DWs, const f32x4 *LFs, int lane) {
#pragma unroll
  for (int t = 0; t < 4; ++t) { rg.cur[t] = bnx_ld32(DWs, t, lane); rg.nxt[t] = bnx_ld32(DWs, 4 + t, lane); }
  rg.la[0] = bnx_ld32(LFs, 0, lane); rg.la[1] = bnx_ld32(LFs, 1, lane);
}
// products of stream tile i on a1, a2 (the caller's initial values); advances the ring
template <int R>
__device__ __forceinline__ void bnx_stream_tile(BnxRing &rg, const f32x4 *__restrict__ DWs, const f32x4 *LFs, int i, int n_tiles, int lane,
                                                const BnxK (&h)[R][2], const BnxK (&hs)[R][2], f32x4 (&a1)[R], f32x4 (&a2)[R]) {
  bnx_u4 fn[4];
  const int i2 = min(i + 2, n_tiles - 1);
#pragma unroll
  for (int t = 0; t < 4; ++t) fn[t] = bnx_ld32(BNX_DWSRC(DWs, LFs) + i2 * 256, t, lane);
#pragma unroll
  for (int T = 0; T < 2; ++T) {
    const bnx_u4 ah = rg.la[0], al = rg.la[1];
    const int nf = min(4 * i + 2 * T + 2, 4 * n_tiles - 2);
    rg.la[0] = bnx_ld32(LFs + nf * 64, 0, lane); rg.la[1] = bnx_ld32(LFs + nf * 64, 1, lane);
    BNF_PIN();
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { a1[rt] = BNX_MFMA32(al, h[rt][T].hi, a1[rt]); a2[rt] = BNX_MFMA32(rg.cur[2 * T + 1], hs[rt][T].hi, a2[rt]); }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { a1[rt] = BNX_MFMA32(ah, h[rt][T].lo, a1[rt]); a2[rt] = BNX_MFMA32(rg.cur[2 * T], hs[rt][T].lo, a2[rt]); }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { a1[rt] = BNX_MFMA32(ah, h[rt][T].hi, a1[rt]); a2[rt] = BNX_MFMA32(rg.cur[2 * T], hs[rt][T].hi, a2[rt]); }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) { rg.cur[t] = rg.nxt[t]; rg.nxt[t] = fn[t]; }
}

// the same products with the caller's perturbation fragments fc (no ring); `mid()` runs behind the first K block's products, which it is
// ordered after (the last layer issues the next pair's requests there: behind the loop top's wait for its own operands, not in front of it)
template <int R, class Mid>
__device__ __forceinline__ void bnx_tile_mm(bnx_u4 (&la)[2], const f32x4 *LFs, int i, int n_tiles, int lane, const BnxK (&h)[R][2], const BnxK (&hs)[R][2],
                                            const bnx_u4 (&fc)[4], f32x4 (&a1)[R], f32x4 (&a2)[R], Mid mid) {
#pragma unroll
  for (int T = 0; T < 2; ++T) {
    const bnx_u4 ah = la[0], al = la[1];
    const int nf = min(4 * i + 2 * T + 2, 4 * n_tiles - 2);
    la[0] = bnx_ld32(LFs + nf * 64, 0, lane); la[1] = bnx_ld32(LFs + nf * 64, 1, lane);
    BNF_PIN();
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { a1[rt] = BNX_MFMA32(al, h[rt][T].hi, a1[rt]); a2[rt] = BNX_MFMA32(fc[2 * T + 1], hs[rt][T].hi, a2[rt]); }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { a1[rt] = BNX_MFMA32(ah, h[rt][T].lo, a1[rt]); a2[rt] = BNX_MFMA32(fc[2 * T], hs[rt][T].lo, a2[rt]); }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { a1[rt] = BNX_MFMA32(ah, h[rt][T].hi, a1[rt]); a2[rt] = BNX_MFMA32(fc[2 * T], hs[rt][T].hi, a2[rt]); }
    if (T == 0) {
      asm volatile("" : "+v"(a1[0]), "+v"(a2[0]) :: "memory");
      mid();
    }
  }
}

// hidden layer 64 -> 64 of g (bnf_hidden): stream tiles i0 .. i0 + 3; h -> hn.  wi / wo: the layer's input- / output-sign words
// (pre-shifted; [0]: features 0..31, [1]: 32..63).
template <int R>
__device__ __forceinline__ void bnx_hidden(BnxRing &rg, const f32x4 *__restrict__ DWs, const f32x4 *LFs, int i0, int n_tiles, const f32x4 *BL, int lane,
                                           int g, const uint32_t (&wi)[R][2], const uint32_t (&wo)[R][2], const BnxK (&h)[R][2], BnxK (&hn)[R][2] BNF_PROF_PARAM) {
  BnxK hs[R][2];
#pragma unroll
  for (int rt = 0; rt < R; ++rt)
#pragma unroll
    for (int T = 0; T < 2; ++T) bnx_flip(h[rt][T], wi[rt][T], hs[rt][T]);
#pragma unroll
  for (int mp = 0; mp < 2; ++mp) {
    f32x4 a1[2][R], a2[2][R];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int mt = 2 * mp + e;
      const f32x4 b = BL[4 * mt + g];
#pragma unroll
      for (int rt = 0; rt < R; ++rt) { a1[e][rt] = b; a2[e][rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      bnx_stream_tile<R>(rg, DWs, LFs, i0 + mt, n_tiles, lane, h, hs, a1[e], a2[e]);
    }
    BNF_T(2);
#pragma unroll
    for (int rt = 0; rt < R; ++rt) bnx_epi_pair(a1[0][rt], a2[0][rt], a1[1][rt], a2[1][rt], wo[rt][mp], hn[rt][mp]);
    BNF_T(7);
  }
}

// ---------------------------------------------------------------------------------------------
// outcome / treatment net  e -> 64 -> 32 -> 8 -> 2 (bnf_head).  Same arguments and sign-word protocol as bnf_head; NORM / SHIFT of the net.
// ---------------------------------------------------------------------------------------------
template <int KS, int R>
__device__ __forceinline__ void bnx_head(const f32x4 *LF, const f32x4 *__restrict__ DW, const f32x4 *BL, const f32x4 *NORM, const int4 *SHIFT, int lane,
                                         int g, const float (&ze)[R][KS], const uint4 (&G)[R][BNF_NG_H], float (&mu)[R], float (&raw)[R]) {
  constexpr int T0 = (KS + 3) / 4;
  const f32x4 *D2 = DW + 4 * T0 * 64;
  // layer 2's perturbation fragments are requested before layer 1, layers 3 / 4's behind it (bnf_head)
  bnx_u4 fd2[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) fd2[mt][t] = bnx_ld32(D2, mt * 4 + t, lane);
  BNF_PIN();
  BnxK h1[R][2], hs1[R][2];      // hs1: formed behind layer 1 from G[.][1].x / .y
  {
    uint32_t w_in[R], wo[R][2];
    uint4 Gn[R];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      w_in[rt] = G[rt][0].x;
      wo[rt][0] = bnf_preshift(G[rt][0].y, g); wo[rt][1] = bnf_preshift(G[rt][0].z, g);
      Gn[rt] = G[rt][1];
    }
    BnxIn<KS> in[R];
    bnx_input<KS, R>(NORM, SHIFT, g, ze, w_in, in);
    bnx_first<KS, R>(LF, DW, BL, lane, g, in, wo, h1);
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { bnx_flip(h1[rt][0], bnf_preshift(Gn[rt].x, g), hs1[rt][0]); bnx_flip(h1[rt][1], bnf_preshift(Gn[rt].y, g), hs1[rt][1]); }
  }
  const bnx_u4 fd3h = bnx_ld32(D2, 8, lane), fd3l = bnx_ld32(D2, 9, lane);
  const bnx_u2 fd4h = bnx_ld16(D2, 10, lane, 0), fd4l = bnx_ld16(D2, 10, lane, 1);
  BNF_PIN();
  // layer 2: 64 -> 32 (one output tile pair)
  const f32x4 *L2 = LF + 4 * T0 * 64;
  BnxK h2[R], hs2[R];
  {
    f32x4 a1[2][R], a2[2][R];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const f32x4 b = BL[4 * (4 + mt) + g];
#pragma unroll
      for (int rt = 0; rt < R; ++rt) { a1[mt][rt] = b; a2[mt][rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int T = 0; T < 2; ++T) {
        const bnx_u4 ah = bnx_ld32(L2, mt * 4 + 2 * T, lane), al = bnx_ld32(L2, mt * 4 + 2 * T + 1, lane);
#pragma unroll
        for (int rt = 0; rt < R; ++rt) { a1[mt][rt] = BNX_MFMA32(al, h1[rt][T].hi, a1[mt][rt]); a2[mt][rt] = BNX_MFMA32(fd2[mt][2 * T + 1], hs1[rt][T].hi, a2[mt][rt]); }
#pragma unroll
        for (int rt = 0; rt < R; ++rt) { a1[mt][rt] = BNX_MFMA32(ah, h1[rt][T].lo, a1[mt][rt]); a2[mt][rt] = BNX_MFMA32(fd2[mt][2 * T], hs1[rt][T].lo, a2[mt][rt]); }
#pragma unroll
        for (int rt = 0; rt < R; ++rt) { a1[mt][rt] = BNX_MFMA32(ah, h1[rt][T].hi, a1[mt][rt]); a2[mt][rt] = BNX_MFMA32(fd2[mt][2 * T], hs1[rt][T].hi, a2[mt][rt]); }
      }
    }
#pragma unroll
    for (int rt = 0; rt < R; ++rt)
    { bnx_epi_pair(a1[0][rt], a2[0][rt], a1[1][rt], a2[1][rt], bnf_preshift(G[rt][1].z, g), h2[rt]); bnx_flip(h2[rt], bnf_preshift(G[rt][1].w, g), hs2[rt]); }
  }
  // layer 3: 32 -> 8, output feature f at lane group f >> 1, register f & 1 (registers 2, 3: zero rows of the packed weights)
  const f32x4 *L3 = L2 + 8 * 64;
  BnxK16 h3[R], hs3[R];
  {
    const bnx_u4 ah = bnx_ld32(L3, 0, lane), al = bnx_ld32(L3, 1, lane);
    const f32x4 b = BL[4 * 6 + g];
    f32x4 a1[R], a2[R];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { a1[rt] = b; a2[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { a1[rt] = BNX_MFMA32(al, h2[rt].hi, a1[rt]); a2[rt] = BNX_MFMA32(fd3l, hs2[rt].hi, a2[rt]); }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { a1[rt] = BNX_MFMA32(ah, h2[rt].lo, a1[rt]); a2[rt] = BNX_MFMA32(fd3h, hs2[rt].lo, a2[rt]); }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { a1[rt] = BNX_MFMA32(ah, h2[rt].hi, a1[rt]); a2[rt] = BNX_MFMA32(fd3h, hs2[rt].hi, a2[rt]); }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      const uint32_t so = G[rt][2].x << (30 - 2 * g), si = G[rt][2].y << (30 - 2 * g);   // bit 2 g + r at position 30 + r
      const float v0 = lrelu_s(fmaf(a2[rt][0], bnf_pm1(so << 1), a1[rt][0])), v1 = lrelu_s(fmaf(a2[rt][1], bnf_pm1(so), a1[rt][1]));
      unsigned hi, lo;
      bnx_split_pair(v0, v1, hi, lo);
      const uint32_t m = (((si << 1) & 0x80000000u) >> 16) | (si & 0x80000000u);
      h3[rt].hi = bnx_u2{hi, 0u}; h3[rt].lo = bnx_u2{lo, 0u};
      hs3[rt].hi = bnx_u2{hi ^ m, 0u}; hs3[rt].lo = bnx_u2{lo ^ m, 0u};
    }
  }
  // layer 4: 8 -> 2, one K = 16 block (slots 4 g + {0, 1}); column o at every 4 g + o
  {
    const bnx_u2 ah = bnx_ld16(L3, 2, lane, 0), al = bnx_ld16(L3, 2, lane, 1);
    const f32x4 b = BL[4 * 7 + g];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      f32x4 a1 = b, a2 = f32x4{0.f, 0.f, 0.f, 0.f};
      a1 = BNX_MFMA16(al, h3[rt].hi, a1); a2 = BNX_MFMA16(fd4l, hs3[rt].hi, a2);
      a1 = BNX_MFMA16(ah, h3[rt].lo, a1); a2 = BNX_MFMA16(fd4h, hs3[rt].lo, a2);
      a1 = BNX_MFMA16(ah, h3[rt].hi, a1); a2 = BNX_MFMA16(fd4h, hs3[rt].hi, a2);
      const uint32_t so = G[rt][2].z;
      mu[rt] = fmaf(a2[0], bnf_pm1(so << 31), a1[0]);
      raw[rt] = fmaf(a2[1], bnf_pm1(so << 30), a1[1]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// log posterior of R row tiles for one state (bnf_logpost_rows for Bayesian nets whose blob fits the LDS)
// ---------------------------------------------------------------------------------------------
template <int KS, int R>
__device__ __forceinline__ void bnx_logpost_rows(const BnfMhArgs &a, const BnfLds &L, int lane, int j, int g, long long blk_lo, const int (&rib)[R],
                                                 const float (&ze)[R][KS], const float (&xr)[R], const float (&yr)[R], const float *dwset, int s,
                                                 const float (&zz)[R], float (&lp)[R] BNF_PROF_PARAM) {
  constexpr int T0 = (KS + 3) / 4;
  const BnfPlan &P = a.pl;
  const int p = P.p;
  const f32x4 *DW = (const f32x4 *)dwset;
  BGM_NO_HOIST();
  BNF_T(0);
  float ssq[R], rawv[R];
  uint4 GH[R][BNF_NG_H];
  {
    BnxK h[R][2];
    const uint4 *SG = a.sg.g + ((long long)s * BNF_NG_G * a.n + blk_lo);
    uint4 Gc[R], Gn[R];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { Gc[rt] = SG[rib[rt]]; Gn[rt] = (SG + a.n)[rib[rt]]; }
    // the generator's stream behind its first layer: hidden layers 1..4 (16 tiles), last layer (NTL tiles)
    const int NTL = P.NTL, n_tiles = 16 + NTL;
    const f32x4 *DWs = DW + P.fgh * 64, *LFs = L.frag + P.fgh * 64;
    BnxRing rg;
    bnx_ring_fill(rg, DWs, LFs, lane);
    // the rows of V of this evaluation's tiles are brought into L2 now (one dword per 128-byte line: lane (j, g) touches lines g and
    // g + 4 of row j of each tile); the last layer's requests, a pair ahead of their use, then find them there
    uint32_t touch[2 * R];
    {
      const float *vb_ = a.v + blk_lo * p;
      const int last_w = p - 1;
#pragma unroll
      for (int rt = 0; rt < R; ++rt)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          touch[2 * rt + c] = __builtin_bit_cast(uint32_t, vb_[rib[rt] * p + min(32 * (g + 4 * c), last_w)]);
    }
    BNF_PIN();
    uint32_t wo[R][2], wi[R][2];
    {
      uint32_t w_in[R];
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        w_in[rt] = Gc[rt].x;
        wo[rt][0] = bnf_preshift(Gc[rt].y, g); wo[rt][1] = bnf_preshift(Gc[rt].z, g);
      }
      BnxIn<KS> in[R];
      bnx_input<KS, R>(L.norm, L.shift, g, ze, w_in, in);
      bnx_first<KS, R>(L.frag + P.fg0 * 64, DW + P.fg0 * 64, L.bias + 4 * P.bg0, lane, g, in, wo, h);
    }
    BNF_T(1);
    // two layers per trip: h -> hb -> h.  A layer's sign group Gn = [in 0..31 | in 32..63 | out 0..31 | out 32..63] was requested one layer
    // ago; the next group is requested now and first touched at the next layer's start
#pragma nounroll          // (fully unrolled: 1.43 against 1.46 ms per iteration at N = 1e6 for 6 KB more code; not taken)
    for (int l = 1; l <= 4; l += 2) {
      BnxK hb[R][2];
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        wi[rt][0] = bnf_preshift(Gn[rt].x, g); wi[rt][1] = bnf_preshift(Gn[rt].y, g);
        wo[rt][0] = bnf_preshift(Gn[rt].z, g); wo[rt][1] = bnf_preshift(Gn[rt].w, g);
      }
#pragma unroll
      for (int rt = 0; rt < R; ++rt) Gn[rt] = (SG + (long long)(l + 1) * a.n)[rib[rt]];
      BNF_PIN();
      bnx_hidden<R>(rg, DWs, LFs, 4 * (l - 1), n_tiles, L.bias + 4 * (P.bgh + 4 * (l - 1)), lane, g, wi, wo, h, hb BNF_PROF_ARG);
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        wi[rt][0] = bnf_preshift(Gn[rt].x, g); wi[rt][1] = bnf_preshift(Gn[rt].y, g);
        wo[rt][0] = bnf_preshift(Gn[rt].z, g); wo[rt][1] = bnf_preshift(Gn[rt].w, g);
      }
#pragma unroll
      for (int rt = 0; rt < R; ++rt) Gn[rt] = (SG + (long long)(l + 2) * a.n)[rib[rt]];
      BNF_PIN();
      bnx_hidden<R>(rg, DWs, LFs, 4 * l, n_tiles, L.bias + 4 * (P.bgh + 4 * l), lane, g, wi, wo, hb, h BNF_PROF_ARG);
    }
    BNF_T(2);
#pragma unroll
    for (int c = 0; c < 2 * R; ++c) asm volatile("" :: "v"(touch[c]));
    BnxK hs[R][2];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { bnx_flip(h[rt][0], bnf_preshift(Gn[rt].x, g), hs[rt][0]); bnx_flip(h[rt][1], bnf_preshift(Gn[rt].y, g), hs[rt][1]); }
    // last layer: 64 -> p + 1 against the data row, in tile PAIRS (the two tiles of a pair share an output-sign word with their bits 16
    // apart: one packed (+-1, +-1) per register serves both); stream tiles 16 .. 16 + NTL - 1.  Software pipeline over pairs: a pair's
    // operands (perturbation fragments of both tiles, data rows, sign word) are requested one pair ahead into the other of two register
    // sets, behind the first products of the pair in flight (the compiler waits for ALL outstanding requests at a loop's head).
    const f32x4 *BLl = L.bias + 4 * P.bgl;
    const uint32_t *GO = a.sg.gout + ((long long)s * a.n + blk_lo) * BNF_GOUT;
    const float *vblk = a.v + blk_lo * p;
    auto load_v = [&](int rt, int mt) __attribute__((always_inline)) -> f32x4 {
#ifdef BNX_ABL_NOV
      const float *vr = vblk + min(16 * mt + 4 * g, p - 4);
#else
      const float *vr = vblk + rib[rt] * p + min(16 * mt + 4 * g, p - 4);
#endif
      return *(const f32x4_u *)vr;
    };
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { ssq[rt] = 0.0f; rawv[rt] = 0.0f; }
    struct PairSet { bnx_u4 fa[4], fb[4]; };
    const int NP = (NTL - 1) >> 1;      // full pairs (2k, 2k + 1), k < NP: none of their columns is the variance column or padding
    // perturbation fragments of pair kp <= NP (kp = NP: the tail's one or two tiles)
    auto request = [&](int kp, PairSet &S) __attribute__((always_inline)) {
      const int ta = 2 * kp, tb = min(2 * kp + 1, NTL - 1);
#pragma unroll
      for (int t = 0; t < 4; ++t) { S.fa[t] = bnx_ld32(BNX_DWSRC(DWs, LFs) + (16 + ta) * 256, t, lane); S.fb[t] = bnx_ld32(BNX_DWSRC(DWs, LFs) + (16 + tb) * 256, t, lane); }
    };
    const uint32_t k8 = bnx_k8(), k3c = bnx_k3c();
    // data rows and sign word of a pair are requested at its start and first touched in its epilogue (the rows were brought into L2 at the
    // start of the evaluation)
    auto pair_body = [&](int k, const PairSet &S, auto next) __attribute__((always_inline)) {
      const int mt = 2 * k;
      f32x4 a1e[R], a2e[R], a1o[R], a2o[R], va[R], vb[R];
      uint32_t w[R];
#pragma unroll
      for (int rt = 0; rt < R; ++rt) { va[rt] = load_v(rt, mt); vb[rt] = load_v(rt, mt + 1); w[rt] = GO[rib[rt] * BNF_GOUT + k]; }
      const f32x4 be = BLl[4 * mt + g], bo = BLl[4 * (mt + 1) + g];
#pragma unroll
      for (int rt = 0; rt < R; ++rt) { a1e[rt] = be; a1o[rt] = bo; a2e[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; a2o[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      bnx_tile_mm<R>(rg.la, LFs, 16 + mt, n_tiles, lane, h, hs, S.fa, a1e, a2e, next);
      bnx_tile_mm<R>(rg.la, LFs, 16 + mt + 1, n_tiles, lane, h, hs, S.fb, a1o, a2o, [] {});
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        const uint32_t wsh = bnf_preshift(w[rt], g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t to = r == 3 ? wsh : wsh << (3 - r);
          const bnx_h2 pm = __builtin_bit_cast(bnx_h2, (to & k8) | k3c);
          const float de = fmaf(a2e[rt][r], (float)pm[0], a1e[rt][r]) - va[rt][r], dq = fmaf(a2o[rt][r], (float)pm[1], a1o[rt][r]) - vb[rt][r];
          ssq[rt] = fmaf(de, de, ssq[rt]);
          ssq[rt] = fmaf(dq, dq, ssq[rt]);
        }
      }
    };
    // single tile of the tail (columns u >= p -- the variance column and the padding -- have no data; a clamped request is shifted back)
    auto tile = [&](int mt, bool last, const f32x4 (&vc)[R], const uint32_t (&wc)[R], const bnx_u4 (&fc)[4]) __attribute__((always_inline)) {
      f32x4 a1[R], a2[R];
      const f32x4 b = BLl[4 * mt + g];
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        a1[rt] = b - vc[rt];
        if (last) {
          const int u0 = 16 * mt + 4 * g, sh = u0 - min(u0, p - 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float vv = sh == 0 ? vc[rt][r] : sh == 1 ? (r < 3 ? vc[rt][r + 1 > 3 ? 3 : r + 1] : 0.f) : sh == 2 ? (r < 2 ? vc[rt][r + 2 > 3 ? 3 : r + 2] : 0.f)
                                   : sh == 3 ? (r < 1 ? vc[rt][3] : 0.f) : 0.f;
            a1[rt][r] = b[r] - (u0 + r < p ? vv : 0.0f);
          }
        }
        a2[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      bnx_tile_mm<R>(rg.la, LFs, 16 + mt, n_tiles, lane, h, hs, fc, a1, a2, [] {});
      const int pos = 16 * (mt & 1);
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        const uint32_t wsh = bnf_preshift(wc[rt], g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = fmaf(a2[rt][r], bnf_sign_rt(wsh, pos + r), a1[rt][r]);
          if (!last) ssq[rt] = fmaf(d, d, ssq[rt]);
          else {
            const int u = 16 * mt + 4 * g + r;
            ssq[rt] = fmaf(u < p ? d : 0.0f, d, ssq[rt]);
            rawv[rt] += (u == p) ? d : 0.0f;
          }
        }
      }
    };
    auto tail = [&](const PairSet &S) __attribute__((always_inline)) {
      f32x4 va[R], vb[R];
      uint32_t w[R];
#pragma unroll
      for (int rt = 0; rt < R; ++rt) { va[rt] = load_v(rt, 2 * NP); vb[rt] = load_v(rt, NTL - 1); w[rt] = GO[rib[rt] * BNF_GOUT + NP]; }
      if (NTL & 1) tile(NTL - 1, true, va, w, S.fa);
      else { tile(NTL - 2, false, va, w, S.fa); tile(NTL - 1, true, vb, w, S.fb); }
    };
    PairSet X, Y;
#pragma unroll
    for (int t = 0; t < 4; ++t) { X.fa[t] = rg.cur[t]; X.fb[t] = rg.nxt[t]; }      // the ring holds stream tiles 16, 17 (an NTL of 1: twice tile 16)
    // the treatment net's sign groups: requested ahead of the tail, first touched behind it (ahead of the whole layer they cost
    // 24 registers through it: measured slower)
    auto request_gh = [&]() __attribute__((always_inline)) {
      const uint4 *SH = a.sg.h + ((long long)s * BNF_NG_H * a.n + blk_lo);
#pragma unroll
      for (int rt = 0; rt < R; ++rt)
#pragma unroll
        for (int kk = 0; kk < BNF_NG_H; ++kk) GH[rt][kk] = (SH + (long long)kk * a.n)[rib[rt]];
    };
    int k = 0;
#pragma nounroll
    for (; k + 2 <= NP; k += 2) {
      pair_body(k, X, [&] { request(k + 1, Y); });
      pair_body(k + 1, Y, [&] { request(k + 2, X); });
    }
    if (k < NP) {
      pair_body(k, X, [&] { request(k + 1, Y); });
      request_gh();
      tail(Y);
    } else { request_gh(); tail(X); }
    BNF_T(3);
  }
  float part[R];
#pragma unroll
  for (int rt = 0; rt < R; ++rt) {
    const float rw = sum_over_g(rawv[rt]);
    const float s2 = (a.sig2_v > 0.0f) ? a.sig2_v : bnf_softplus(rw) + BGM_EPS;
    part[rt] = -(ssq[rt] * fast_rcp(2.0f * s2) + 0.5f * zz[rt]);
    lp[rt] = -0.5f * (float)p * fast_log(s2);
  }
  // ---- h (treatment) and f (outcome)
  {
    float mu[R], raw[R];
    uint4 G[R][BNF_NG_H];
    {
      const uint4 *SF = a.sg.f + ((long long)s * BNF_NG_H * a.n + blk_lo);
#pragma unroll
      for (int rt = 0; rt < R; ++rt)
#pragma unroll
        for (int k = 0; k < BNF_NG_H; ++k) G[rt][k] = (SF + (long long)k * a.n)[rib[rt]];
    }
    BGM_NO_HOIST();
    bnx_head<KS, R>(L.frag + P.fh * 64, DW + P.fh * 64, L.bias + 4 * P.bh, L.norm + 1 * T0 * 8, L.shift + 1 * T0 * 4, lane, g, ze, GH, mu, raw);
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      const float m_ = mu[rt];
      if (P.binary) lp[rt] -= fmaxf(m_, 0.0f) - m_ * xr[rt] + bnf_softplus(-fabsf(m_));
      else {
        const float s2 = (a.sig2_x > 0.0f) ? a.sig2_x : bnf_softplus(raw[rt]) + BGM_EPS, d = xr[rt] - m_;
        lp[rt] -= d * d * fast_rcp(2.0f * s2) + 0.5f * fast_log(s2);
      }
    }
    BNF_T(4);
    BGM_NO_HOIST();
    bnx_head<KS, R>(L.frag + P.ff * 64, DW + P.ff * 64, L.bias + 4 * P.bf, L.norm + 2 * T0 * 8, L.shift + 2 * T0 * 4, lane, g, ze, G, mu, raw);
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      const float s2 = (a.sig2_y > 0.0f) ? a.sig2_y : bnf_softplus(raw[rt]) + BGM_EPS, d = yr[rt] - mu[rt];
      lp[rt] -= d * d * fast_rcp(2.0f * s2) + 0.5f * fast_log(s2);
    }
  }
#pragma unroll
  for (int rt = 0; rt < R; ++rt) lp[rt] += sum_over_g(part[rt]);
  BNF_T(5);
}

// ---------------------------------------------------------------------------------------------
// packing and perturbations in the split layout
// ---------------------------------------------------------------------------------------------
// position word of the second table: position in fp16 units (27 bits) | K = 16 block (bit 27) | copies - 1 (top nibble); the lo half of
// an element sits 512 (K = 32) or 256 (K = 16) fp16 units behind its hi half; copies (the replicated columns of the 8 -> 2 layers) are
// 16 fp16 units apart
__device__ __forceinline__ void bnx_store(bnx_h *dst, int pe, float v) {
  const int pos = pe & 0x07FFFFFF, lo_off = (pe & 0x08000000) ? 256 : 512, rep = (pe >> 28) & 15;
  const float c = fminf(fmaxf(v, -65504.0f), 65504.0f);
  const bnx_h hi = (bnx_h)c, lo = (bnx_h)(c - (float)hi);
  for (int r = 0; r <= rep; ++r) { dst[pos + 16 * r] = hi; dst[pos + lo_off + 16 * r] = lo; }
}

// blobx <- blob: the fragments re-encoded (the fp32 blob already carries the 0.6 of the one-instruction LeakyReLU), the tail copied
struct BnxPackArgs {
  const float *blob; float *blobx;
  const BnfWElem *w; int n_w;
  const int *posx;
  int frag_floats, blob_floats;
};
static __global__ void bnx_pack_kernel(BnxPackArgs a) {
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x, str = gridDim.x * blockDim.x;
  for (int i = i0; i < a.n_w; i += str) bnx_store((bnx_h *)a.blobx, a.posx[i], a.blob[a.w[i].pos]);
  for (int i = a.frag_floats + i0; i < a.blob_floats; i += str) a.blobx[i] = a.blob[i];
}

// bnf_noise_kernel writing hi / lo fp16 at the positions of the second table (same Philox calls, same sigma fragments)
struct BnxNoiseArgs { BnfNoiseArgs n; const int *posx; };
static __global__ __launch_bounds__(256) void bnx_noise_kernel(BnxNoiseArgs b) {
  const BnfNoiseArgs &a = b.n;
  const int set = blockIdx.y, blk = set / a.n_states, s = set - blk * a.n_states;
  const uint32_t k1 = a.k1 + (uint32_t)(a.block0 + blk), stream = a.stream0 + (uint32_t)s;
  bnx_h *dw = (bnx_h *)(a.dw + (long long)set * a.set_floats);
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < a.n_calls; c += gridDim.x * blockDim.x) {
    int d = 0;
#pragma unroll
    for (int k = 1; k < 14; ++k) d += (k < a.n_lay && c >= a.lay[k].c_base) ? 1 : 0;
    const BnfLayerDesc L = a.lay[d];
    const int i = c - L.c_base;
    const f32x4 z = box_muller4(philox4x32_10((uint32_t)i, (uint32_t)L.l | ((uint32_t)L.net_id << 16), stream, BNN_TAG_EPS, a.k0, k1));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = 4 * i + u;
      if (idx < L.cnt) bnx_store(dw, b.posx[L.e_base + idx], a.sf[a.npos[L.e_base + idx] & 0x0FFFFFFF] * z[u]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Metropolis-Hastings iteration around the MODE 3 evaluation (bnf_mh_kernel's MODE 1 in three launches): the proposal and the accept
// step of metropolis_hastings_sampler (causalbgm/base.py:860-871) as row-wise kernels with MODE 1's Philox calls and expressions, so
// that chains equal the fused kernel's given equal log posteriors.  The evaluation kernel then carries neither the two states nor the
// accept bookkeeping in registers.
// ---------------------------------------------------------------------------------------------
struct BnxMhStepArgs {
  float *z;                   // [n x q] current states (propose with init: written; accept: updated)
  float *zprop;               // [n x q]
  const float *lp;            // accept: [2][n] log posterior of the proposals | of the current states
  long long n, row_base;
  int q, bs, it, init;
  float q_sd;
  const float *q_sd_blocks;
  uint32_t k0, k1;
  unsigned *acc_count, *acc_blocks;
};
// one thread per (row, Philox call c = gg + 4 sb): features 16 sb + 4 r + gg, r = 0..3
static __global__ __launch_bounds__(256) void bnx_propose_kernel(BnxMhStepArgs a) {
  const int calls = 4 * ((a.q + 15) >> 4);
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = t / calls;
  if (row >= a.n) return;
  const int c = (int)(t - row * calls), gg = c & 3, sb = c >> 2;
  const uint32_t rid = (uint32_t)(a.row_base + row);
  const float sd = a.q_sd_blocks ? a.q_sd_blocks[row / a.bs] : a.q_sd;
  float *zr = a.z + row * a.q, *zp = a.zprop + row * a.q;
  f32x4 zc;
  if (a.init) {
    zc = box_muller4(philox4x32_10(rid, 0u, (uint32_t)c, TAG_INIT, a.k0, a.k1));
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int f = 16 * sb + 4 * r + gg; if (f < a.q) zr[f] = zc[r]; }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int f = 16 * sb + 4 * r + gg; zc[r] = f < a.q ? zr[f] : 0.0f; }
  }
  const f32x4 e = box_muller4(philox4x32_10(rid, (uint32_t)a.it, (uint32_t)c, TAG_PROP, a.k0, a.k1));
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int f = 16 * sb + 4 * r + gg; if (f < a.q) zp[f] = fmaf(sd, e[r], zc[r]); }
}
// one thread per row
static __global__ __launch_bounds__(256) void bnx_accept_kernel(BnxMhStepArgs a) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool in = row < a.n;
  bool acc = false;
  int blk = 0;
  if (in) {
    const uint4 w4 = philox4x32_10((uint32_t)(a.row_base + row), (uint32_t)a.it >> 2, 0u, TAG_ACC, a.k0, a.k1);
    const int it = a.it;
    const unsigned w = (it & 2) ? ((it & 1) ? w4.w : w4.z) : ((it & 1) ? w4.y : w4.x);
    acc = u01_open(w) < fast_exp(fminf(a.lp[row] - a.lp[a.n + row], 0.0f));
    blk = (int)(row / a.bs);
    if (acc) {
      const float *zp = a.zprop + row * a.q;
      float *zr = a.z + row * a.q;
      for (int f = 0; f < a.q; ++f) zr[f] = zp[f];
    }
  }
  if (a.acc_count || a.acc_blocks) {
    // one atomic per block of rows the wave's accepted rows lie in (64 consecutive rows: one block, a few when bs is small)
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(acc);
    if (a.acc_count && lane == 0 && todo) atomicAdd(a.acc_count, (unsigned)__popcll(todo));
    while (a.acc_blocks && todo) {
      const int l = __builtin_ctzll(todo), b = __shfl(blk, l);
      const unsigned long long same = __ballot(acc && blk == b);
      if (lane == l) atomicAdd(&a.acc_blocks[b], (unsigned)__popcll(same));
      todo &= ~same;
    }
  }
}
