// causal_bx3_kernels.h -- split-precision ("bf16 x 3" / "f16 x 3") variant of the CausalBGM sampling kernels (gfx950), opt-in
// (`bgm_causal_set_precision(h, 1 | 2)`, params['mh_precision'] = 'bf16x3' | 'f16x3'); fp32 (causal_kernels.h) stays the default.
// Two 16-bit operand formats, one source: bf16 (hi + lo = 16 mantissa bits, fp32 range) and fp16 (22 bits -- the log posterior
// is as close to float64 as the fp32 kernel's -- weights beyond 65504 are clamped by the packer, an activation beyond 65504 overflows); same instruction count, same matrix rate.
//
// replaces (src/bayesgm/models/causalbgm/base.py): get_log_posterior :765-817, metropolis_hastings_sampler :820-904,
// infer_from_latent_posterior :671-763 -- the same algorithm, RNG streams and data layout as causal_kernels.h; only the dense
// contractions change: every weight matrix and every activation is split into two bf16 numbers, x = x_hi + x_lo, and
//     W h  ~=  W_hi h_hi + W_hi h_lo + W_lo h_hi          (fp32 accumulation, the W_lo h_lo term ~2^-16 |W h| is dropped)
// on v_mfma_f32_16x16x32_bf16 (K = 32 per instruction at ~17 cycles, against K = 4 at 32 cycles for the fp32 MFMA): a
// 64 -> 64 layer is 24 matrix instructions instead of 64.  Relative error of one layer against float64: ~6e-6 (fp32: 2.4e-7).
//
// Layout.  M = output feature, N = chain (16 rows per wave), K = input feature, as in the fp32 kernels, so a layer's
// accumulators (lane (j, g), tile t, register r = feature 16 t + 4 g + r of row j) are again the next layer's B operands:
// a K block of 32 is the pair of tiles (2T, 2T+1), lane group g supplies k-slots 8 g + u, u = 4 s + r  <->  feature
// 16 (2T + s) + 4 g + r, and the weights are packed in that K order.  An odd last tile (16 inputs: the first layers at
// z_dims [1,1,1,7], the 8 -> 2 output layers of f / h) uses v_mfma_f32_16x16x16_bf16, k-slot 4 g + r.
// Blob (LDS resident, bytes): per layer the A fragments [tile][K block][hi | lo][64 lanes] of 16 B (8 B for a K = 16 block),
// then the fp32 biases / x-row of f in accumulator order (float offsets).  Weights behind a LeakyReLU carry the factor 0.6 of
// the one-instruction activation lrelu_s, as in the fp32 sampling blob.
// The unit is written once for a 16-bit operand format and compiled per format: the including file defines BX_NS (namespace) and
// BX_F16 (0: bf16, v_mfma_f32_16x16x32_bf16; 1: fp16, v_mfma_f32_16x16x32_f16) before each inclusion.
#include "causal_kernels.h"

#ifndef BX_COMMON_DEFINED
#define BX_COMMON_DEFINED
struct BxMeta {                 // run-time part of the model description
  int q, p, sig_pc, binary, n_gh;
  float sig2_v, sig2_x, sig2_y;
  int total_bytes;
};

__host__ __device__ constexpr int bx_layer_bytes(int KT, int NT) { return NT * ((KT / 2) * 2048 + (KT & 1) * 1024); }

// Blob layout of a compiled shape (KT1 first-layer input tiles, NTL output tiles of g's last layer): every offset except the
// hidden stack of g (n_gh layers, placed last) is a compile-time constant, so the kernels carry no offset table in scalar
// registers.  Float vectors first (float offsets), then the packed weights (byte offsets), then g's hidden biases and weights.
template <int KT1, int NTL>
struct BxLayout {
  // fp32 vectors, float offsets
  static constexpr int b1g = 0, b1f = 64, b1h = 128, bgl = 192, bf2 = bgl + 16 * NTL, bf3 = bf2 + 32, bf4 = bf3 + 16, bh2 = bf4 + 16,
                       bh3 = bh2 + 32, bh4 = bh3 + 16, wxf = bh4 + 16, n_floats = wxf + 64;
  // packed weights, byte offsets
  static constexpr int w1g = 4 * n_floats, w1f = w1g + bx_layer_bytes(KT1, 4), w1h = w1f + bx_layer_bytes(KT1, 4),
                       wgl = w1h + bx_layer_bytes(KT1, 4), wf2 = wgl + bx_layer_bytes(4, NTL), wf3 = wf2 + bx_layer_bytes(4, 2),
                       wf4 = wf3 + bx_layer_bytes(2, 1), wh2 = wf4 + bx_layer_bytes(1, 1), wh3 = wh2 + bx_layer_bytes(4, 2),
                       wh4 = wh3 + bx_layer_bytes(2, 1), fixed_bytes = wh4 + bx_layer_bytes(1, 1);
  // hidden stack of g: n_gh x 64 biases (float offset bg), then n_gh layers of 16 KiB (byte offset wg(n_gh))
  static constexpr int bg = fixed_bytes / 4;
  __host__ __device__ static constexpr int wg(int n_gh) { return fixed_bytes + 256 * n_gh; }
  __host__ __device__ static constexpr int total(int n_gh) { return fixed_bytes + 256 * n_gh + n_gh * bx_layer_bytes(4, 4); }
};

struct CausalBxKArgs {
  CausalMhKArgs a;        // the fp32 kernel's arguments (a.blob / a.m unused)
  const unsigned char *bblob;
  BxMeta bx;
};

#endif  // BX_COMMON_DEFINED

namespace BX_NS {
#if BX_F16
typedef _Float16 bx_elem;
#define BX_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define BX_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 bx_elem;
#define BX_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define BX_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0)
#endif
typedef bx_elem bx_x8 __attribute__((ext_vector_type(8)));
typedef bx_elem bx_x4 __attribute__((ext_vector_type(4)));
typedef bx_elem bx_x2 __attribute__((ext_vector_type(2)));

// ---- fragment reads.  The blob is larger than the 16-bit offset field of ds_read, and hipcc folds a layer's constant blob offset into
// every read's own address (one v_add_u32 per ds_read_b128 in the ISA).  The per-lane base of a layer is therefore made opaque once
// (LDS byte offset in a VGPR) and every fragment of the layer is read at base + immediate.
typedef const __attribute__((address_space(3))) unsigned char *bx_lds_ptr;
__device__ __forceinline__ bx_lds_ptr bx_frag_base(const unsigned char *w, int lane_bytes) {
  unsigned off = (unsigned)(unsigned long long)(bx_lds_ptr)w + (unsigned)lane_bytes;
  asm volatile("" : "+v"(off));
  return (bx_lds_ptr)(unsigned long long)off;
}
__device__ __forceinline__ bx_x8 bx_ld8(bx_lds_ptr b, int off) {
  return *reinterpret_cast<const __attribute__((address_space(3))) bx_x8 *>(b + off);
}
__device__ __forceinline__ bx_x4 bx_ld4(bx_lds_ptr b, int off) {
  return *reinterpret_cast<const __attribute__((address_space(3))) bx_x4 *>(b + off);
}

// ---- activation split: 8 (or 4) fp32 values -> bf16 hi / lo -------------------------------------------------
// One pair per step: v_cvt_pk_bf16_f32 (hi pair, RNE), the two hi values back in fp32 as shift / mask of the packed word, two exact
// subtractions, v_cvt_pk_bf16_f32 (lo pair): 3 VALU instructions per element.
typedef float bx_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned bx_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned bx_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bx_split_pair(float a, float b, unsigned &hi, unsigned &lo) {
  const bx_x2 h = __builtin_convertvector(bx_f32x2{a, b}, bx_x2);
  hi = __builtin_bit_cast(unsigned, h);
#if BX_F16
  // (v_fma_mix_f32 would form a - float(hi) in one instruction, but it reads a subnormal fp16 operand as zero whatever the mode
  // register says: |a| < 2^-14 came out counted twice, 6e-3 on a sensitive row of the log-posterior test.  v_cvt_f32_f16 does not.)
  const float la = a - (float)h[0], lb = b - (float)h[1];
  const bx_x2 l = __builtin_convertvector(bx_f32x2{la, lb}, bx_x2);
#else
  const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
  const bx_x2 l = __builtin_convertvector(bx_f32x2{a - ha, b - hb}, bx_x2);
#endif
  lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void bx_split8(const f32x4 &a, const f32x4 &b, bx_x8 &hi, bx_x8 &lo) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  bx_split_pair(a[0], a[1], h0, l0);
  bx_split_pair(a[2], a[3], h1, l1);
  bx_split_pair(b[0], b[1], h2, l2);
  bx_split_pair(b[2], b[3], h3, l3);
  hi = __builtin_bit_cast(bx_x8, bx_u32x4{h0, h1, h2, h3});
  lo = __builtin_bit_cast(bx_x8, bx_u32x4{l0, l1, l2, l3});
}
__device__ __forceinline__ void bx_split4(const f32x4 &a, bx_x4 &hi, bx_x4 &lo) {
  unsigned h0, h1, l0, l1;
  bx_split_pair(a[0], a[1], h0, l0);
  bx_split_pair(a[2], a[3], h1, l1);
  hi = __builtin_bit_cast(bx_x4, bx_u32x2{h0, h1});
  lo = __builtin_bit_cast(bx_x4, bx_u32x2{l0, l1});
}

// acc[NT] += W^T in   for one layer; `w` = LDS byte address of the layer's fragments, `in` = KT activated input tiles.
// The input is split once; the output tiles are processed in groups of up to four (A fragments of a group: 32 registers), and
// per K block the three products of a group are issued tile-major per product, so that consecutive MFMAs write different
// accumulators.
// split form of KT activated input tiles: K = 32 blocks (hi, lo) and, for odd KT, the trailing K = 16 block
template <int KT>
struct BxIn {
  static constexpr int NK32 = KT / 2, K16 = KT & 1;
  bx_x8 bh[NK32 > 0 ? NK32 : 1], bl[NK32 > 0 ? NK32 : 1];
  bx_x4 ch, cl;
  __device__ __forceinline__ void split(const f32x4 (&in)[KT]) {
#pragma unroll
    for (int T = 0; T < NK32; ++T) bx_split8(in[2 * T], in[2 * T + 1], bh[T], bl[T]);
    if constexpr (K16) bx_split4(in[KT - 1], ch, cl);
  }
};

template <int KT, int NT>
__device__ __forceinline__ void bx_dense_mm(const unsigned char *w, int lane, const BxIn<KT> &b, f32x4 (&acc)[NT]) {
  constexpr int NK32 = KT / 2, K16 = KT & 1;
  constexpr int TILE_BYTES = NK32 * 2048 + K16 * 1024;
  const bx_lds_ptr w16 = bx_frag_base(w, lane * 16);
  const bx_lds_ptr w8 = K16 ? bx_frag_base(w + NK32 * 2048, lane * 8) : w16;
#pragma unroll
  for (int m0 = 0; m0 < NT; m0 += 4) {
    constexpr int GSMAX = 4;
    const int gs = (NT - m0 < GSMAX) ? NT - m0 : GSMAX;      // compile-time after unrolling
#pragma unroll
    for (int T = 0; T < NK32; ++T) {
      bx_x8 ah[GSMAX], al[GSMAX];
#pragma unroll
      for (int u = 0; u < GSMAX; ++u)
        if (u < gs) {
          ah[u] = bx_ld8(w16, (m0 + u) * TILE_BYTES + T * 2048);
          al[u] = bx_ld8(w16, (m0 + u) * TILE_BYTES + T * 2048 + 1024);
        }
#pragma unroll
      for (int u = 0; u < GSMAX; ++u) if (u < gs) acc[m0 + u] = BX_MFMA32(al[u], b.bh[T], acc[m0 + u]);
#pragma unroll
      for (int u = 0; u < GSMAX; ++u) if (u < gs) acc[m0 + u] = BX_MFMA32(ah[u], b.bl[T], acc[m0 + u]);
#pragma unroll
      for (int u = 0; u < GSMAX; ++u) if (u < gs) acc[m0 + u] = BX_MFMA32(ah[u], b.bh[T], acc[m0 + u]);
    }
    if constexpr (K16) {
      bx_x4 ah[GSMAX], al[GSMAX];
#pragma unroll
      for (int u = 0; u < GSMAX; ++u)
        if (u < gs) {
          ah[u] = bx_ld4(w8, (m0 + u) * TILE_BYTES);
          al[u] = bx_ld4(w8, (m0 + u) * TILE_BYTES + 512);
        }
#pragma unroll
      for (int u = 0; u < GSMAX; ++u) if (u < gs) acc[m0 + u] = BX_MFMA16(al[u], b.ch, acc[m0 + u]);
#pragma unroll
      for (int u = 0; u < GSMAX; ++u) if (u < gs) acc[m0 + u] = BX_MFMA16(ah[u], b.cl, acc[m0 + u]);
#pragma unroll
      for (int u = 0; u < GSMAX; ++u) if (u < gs) acc[m0 + u] = BX_MFMA16(ah[u], b.ch, acc[m0 + u]);
    }
  }
}

template <int KT, int NT>
__device__ __forceinline__ void bx_dense_acc(const unsigned char *w, int lane, const f32x4 (&in)[KT], f32x4 (&acc)[NT]) {
  BxIn<KT> b;
  b.split(in);
  bx_dense_mm<KT, NT>(w, lane, b, acc);
}

template <int NT>
__device__ __forceinline__ void bx_bias(const float *ldsf, int boff, int g, f32x4 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = *reinterpret_cast<const f32x4 *>(ldsf + boff + 16 * t + 4 * g);
}
template <int NT>
__device__ __forceinline__ void bx_lrelu(f32x4 (&a)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) a[t][r] = lrelu_s(a[t][r]);
}

// Two layers of the same shape (the f and h nets) in lock step: their products are interleaved so that twice as many independent
// accumulators are in flight (alone, the narrow tail layers are chains of dependent MFMAs on one or two accumulators).  NT <= 4.
template <int KT, int NT>
__device__ __forceinline__ void bx_dense_mm2(const unsigned char *wa, const unsigned char *wb, int lane, const BxIn<KT> &ba,
                                             const BxIn<KT> &bb, f32x4 (&acca)[NT], f32x4 (&accb)[NT]) {
  static_assert(NT <= 4, "bx_dense_mm2: one tile group");
  constexpr int NK32 = KT / 2, K16 = KT & 1;
  constexpr int TILE_BYTES = NK32 * 2048 + K16 * 1024;
  const bx_lds_ptr wa16 = bx_frag_base(wa, lane * 16), wb16 = bx_frag_base(wb, lane * 16);
  const bx_lds_ptr wa8 = K16 ? bx_frag_base(wa + NK32 * 2048, lane * 8) : wa16, wb8 = K16 ? bx_frag_base(wb + NK32 * 2048, lane * 8) : wb16;
#pragma unroll
  for (int T = 0; T < NK32; ++T) {
    bx_x8 ah[NT], al[NT], bh[NT], bl[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      ah[u] = bx_ld8(wa16, u * TILE_BYTES + T * 2048); al[u] = bx_ld8(wa16, u * TILE_BYTES + T * 2048 + 1024);
      bh[u] = bx_ld8(wb16, u * TILE_BYTES + T * 2048); bl[u] = bx_ld8(wb16, u * TILE_BYTES + T * 2048 + 1024);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      acca[u] = BX_MFMA32(al[u], ba.bh[T], acca[u]);
      accb[u] = BX_MFMA32(bl[u], bb.bh[T], accb[u]);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      acca[u] = BX_MFMA32(ah[u], ba.bl[T], acca[u]);
      accb[u] = BX_MFMA32(bh[u], bb.bl[T], accb[u]);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      acca[u] = BX_MFMA32(ah[u], ba.bh[T], acca[u]);
      accb[u] = BX_MFMA32(bh[u], bb.bh[T], accb[u]);
    }
  }
  if constexpr (K16) {
    bx_x4 ah[NT], al[NT], bh[NT], bl[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      ah[u] = bx_ld4(wa8, u * TILE_BYTES); al[u] = bx_ld4(wa8, u * TILE_BYTES + 512);
      bh[u] = bx_ld4(wb8, u * TILE_BYTES); bl[u] = bx_ld4(wb8, u * TILE_BYTES + 512);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      acca[u] = BX_MFMA16(al[u], ba.ch, acca[u]);
      accb[u] = BX_MFMA16(bl[u], bb.ch, accb[u]);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      acca[u] = BX_MFMA16(ah[u], ba.cl, acca[u]);
      accb[u] = BX_MFMA16(bh[u], bb.cl, accb[u]);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      acca[u] = BX_MFMA16(ah[u], ba.ch, acca[u]);
      accb[u] = BX_MFMA16(bh[u], bb.ch, accb[u]);
    }
  }
}

// f and h from the shared (split) extended input: first layers and tails 64 -> 32 -> 8 -> 2 in lock step; (mu, s) of both nets
// are valid in every lane (replicated output columns)
template <int KT1, class L>
__device__ __forceinline__ void bx_fh(const unsigned char *lds, const float *ldsf, int lane, int g, const BxIn<KT1> &zsplit, float &mu_y,
                                      float &sr_y, float &mu_x, float &sr_x) {
  f32x4 f1[4], h1[4];
  bx_bias<4>(ldsf, L::b1f, g, f1);
  bx_bias<4>(ldsf, L::b1h, g, h1);
  bx_dense_mm2<KT1, 4>(lds + L::w1f, lds + L::w1h, lane, zsplit, zsplit, f1, h1);
  bx_lrelu<4>(f1);
  bx_lrelu<4>(h1);
  BxIn<4> sf1, sh1;
  sf1.split(f1); sh1.split(h1);
  f32x4 f2[2], h2[2];
  bx_bias<2>(ldsf, L::bf2, g, f2);
  bx_bias<2>(ldsf, L::bh2, g, h2);
  bx_dense_mm2<4, 2>(lds + L::wf2, lds + L::wh2, lane, sf1, sh1, f2, h2);
  bx_lrelu<2>(f2);
  bx_lrelu<2>(h2);
  BxIn<2> sf2, sh2;
  sf2.split(f2); sh2.split(h2);
  f32x4 f3[1], h3[1];
  bx_bias<1>(ldsf, L::bf3, g, f3);
  bx_bias<1>(ldsf, L::bh3, g, h3);
  bx_dense_mm2<2, 1>(lds + L::wf3, lds + L::wh3, lane, sf2, sh2, f3, h3);
  bx_lrelu<1>(f3);
  bx_lrelu<1>(h3);
  BxIn<1> sf3, sh3;
  sf3.split(f3); sh3.split(h3);
  f32x4 f4[1], h4[1];
  bx_bias<1>(ldsf, L::bf4, g, f4);
  bx_bias<1>(ldsf, L::bh4, g, h4);
  bx_dense_mm2<1, 1>(lds + L::wf4, lds + L::wh4, lane, sf3, sh3, f4, h4);
  mu_y = f4[0][0]; sr_y = f4[0][1];
  mu_x = h4[0][0]; sr_x = h4[0][1];
}

// log p(z | x, y, v) of the wave's 16 chains (causal_logp of causal_kernels.h in split precision)
template <int KT1, int NTL>
__device__ __forceinline__ float causal_logp_bx3(const unsigned char *lds, const BxMeta &m, int lane, int g, int j,
                                                 const f32x4 (&zin)[KT1], const f32x4 (&vreg)[NTL], float xr, float yr) {
  using L = BxLayout<KT1, NTL>;
  const float *ldsf = reinterpret_cast<const float *>(lds);
  BxIn<KT1> zsplit;           // the extended input [z, x, 0 ...] feeds the first layers of g, f and h: split once
  zsplit.split(zin);
  float ssq = 0.0f, sraw_v = 0.0f;
  {
    f32x4 h[4];
    bx_bias<4>(ldsf, L::b1g, g, h);
    bx_dense_mm<KT1, 4>(lds + L::w1g, lane, zsplit, h);
    bx_lrelu<4>(h);
    for (int l = 0; l < m.n_gh; ++l) {
      BGM_NO_HOIST();
      f32x4 h2[4];
      bx_bias<4>(ldsf, L::bg + 64 * l, g, h2);
      bx_dense_acc<4, 4>(lds + L::wg(m.n_gh) + l * bx_layer_bytes(4, 4), lane, h, h2);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[t][r] = lrelu_s(h2[t][r]);
    }
    // last layer on top of (bias - v): mu - v per covariate, the variance column at element sig_pc of the last tile
    f32x4 acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) acc[t] = vreg[t];
    bx_dense_acc<4, NTL>(lds + L::wgl, lane, h, acc);
    const int sig_r = m.sig_pc - 4 * g;
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float d = acc[t][r];
        if (t == NTL - 1) {
          const bool is_sig = (r == sig_r);
          sraw_v = is_sig ? d : sraw_v;
          d = is_sig ? 0.0f : d;
        }
        ssq = fmaf(d, d, ssq);
      }
    sraw_v = __shfl(sraw_v, j + 16 * (m.sig_pc >> 2));
  }
  float mu_y, sr_y, mu_x, sr_x;
  bx_fh<KT1, L>(lds, ldsf, lane, g, zsplit, mu_y, sr_y, mu_x, sr_x);
  float zsq = 0.0f;
#pragma unroll
  for (int t = 0; t < KT1; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float zz = zin[t][r];
      zsq = (16 * t + 4 * r + g < m.q) ? fmaf(zz, zz, zsq) : zsq;
    }
  const float s2v = (m.sig2_v > 0.0f) ? m.sig2_v : softplus_f(sraw_v) + BGM_EPS;
  float loss_x;
  if (m.binary) {
    const float l = mu_x;
    const float e = fast_exp(-fabsf(l));
    loss_x = vmax(l, 0.0f) - l * xr + ((e < 2.44140625e-4f) ? e * (1.0f - 0.5f * e) : fast_log(1.0f + e));
  } else {
    const float s2x = (m.sig2_x > 0.0f) ? m.sig2_x : softplus_f(sr_x) + BGM_EPS;
    const float dx = xr - mu_x;
    loss_x = 0.5f * (dx * dx * fast_rcp(s2x) + fast_log(s2x));
  }
  const float s2y = (m.sig2_y > 0.0f) ? m.sig2_y : softplus_f(sr_y) + BGM_EPS;
  const float dy = yr - mu_y;
  const float loss_y = 0.5f * (dy * dy * fast_rcp(s2y) + fast_log(s2y));
  float part = 0.5f * (ssq * fast_rcp(s2v) + zsq);
  part += (g == 0) ? (loss_x + loss_y) : 0.0f;
  return -(sum_over_g(part) + 0.5f * (float)m.p * fast_log(s2v));
}

// vreg = (bias of g's last layer) - (V row), accumulator layout (the bx3 blob keeps its fp32 vectors at float offsets)
template <int NTL>
__device__ __forceinline__ void bx_load_v(const float *v, const float *bl, long long n, int p, long long row0, int j, int g,
                                          f32x4 (&vreg)[NTL]) {
  long long row = row0 + j;
  row = row < n ? row : n - 1;
  const float *vr = v + row * (long long)p;
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * t + 4 * g + r;
      vreg[t][r] = bl[c] - ((c < p) ? vr[c] : 0.0f);
    }
}

__device__ __forceinline__ void bx_lds_fill(unsigned char *lds, const unsigned char *blob, int total_bytes) {
  const f32x4 *src = reinterpret_cast<const f32x4 *>(blob);
  f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
  for (int i = threadIdx.x; i < total_bytes / 16; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
}

// ---------------------------------------------------------------------------
// get_log_posterior for n rows
// ---------------------------------------------------------------------------
template <int KT1, int NTL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void causal_logpost_bx3_kernel(const unsigned char *blob, BxMeta m, const float *x,
                                                                        const float *y, const float *v, const float *z,
                                                                        long long n, float *out, const int *seg, const float *prior_tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bx_lds[];
  using L = BxLayout<KT1, NTL>;
  bx_lds_fill(bx_lds, blob, m.total_bytes);
  const float *ldsf = reinterpret_cast<const float *>(bx_lds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const long long n_tiles = (n + 15) / 16;
  for (long long tile = (long long)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (long long)gridDim.x * WAVES) {
    BGM_NO_HOIST();
    const long long row0 = tile * 16;
    long long row = row0 + j;
    row = row < n ? row : n - 1;
    const float xr[1] = {x[row]}, yr = y[row];
    f32x4 vreg[NTL];
    bx_load_v<NTL>(v, ldsf + L::bgl, n, m.p, row0, j, g, vreg);
    f32x4 zin[1][KT1];
    load_z_rows<KT1, 1>(z, n, m.q, row0, j, g, xr, zin);
    float lp = causal_logp_bx3<KT1, NTL>(bx_lds, m, lane, g, j, zin[0], vreg, xr[0], yr);
    if (seg != nullptr) {      // conditional latent prior (IdentifiableCausalBGM): the row's table entry replaces |z|^2 / 2, in fp32 (PriorRow)
      PriorRow<KT1> pr;
      pr.load(seg, prior_tab, row, m.q, g);
      lp += pr.correction(zin[0], m.q, g);
    }
    if (g == 0 && row0 + j < n) out[row0 + j] = lp;
  }
}

// ---------------------------------------------------------------------------
// infer_from_latent_posterior for the wave's 16 states (causal_effects of causal_kernels.h in split precision):
// f's first layer once at x = 0 (fp32 accumulators), doses as fp32 rank-1 updates, four doses per pass sharing the A
// fragments, lane group g finishing dose e = g of the pass; same noise plan (Philox calls in groups of four).
// ---------------------------------------------------------------------------
template <int KT1, int NTL, int EFFECT, bool CACHE = false>
__device__ __forceinline__ void causal_effects_bx3(const unsigned char *lds, const BxMeta &m, int lane, int g, int j,
                                                   const f32x4 (&zs)[KT1], unsigned rowid, bool valid, long long row, long long n,
                                                   unsigned it, long long d, int n_keep, int sample_y, int n_doses,
                                                   const float *x_values, float *adrf_slot, float *ite, unsigned k0, unsigned k1,
                                                   float2 *cache = nullptr) {
  BGM_NO_HOIST();
  using L = BxLayout<KT1, NTL>;
  const float *ldsf = reinterpret_cast<const float *>(lds);
  f32x4 z0in[KT1];
#pragma unroll
  for (int t = 0; t < KT1; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) z0in[t][r] = (16 * t + 4 * r + g == m.q) ? 0.0f : zs[t][r];
  f32x4 base[4];
  bx_bias<4>(ldsf, L::b1f, g, base);
  bx_dense_acc<KT1, 4>(lds + L::w1f, lane, z0in, base);
  f32x4 wx[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) wx[t] = *reinterpret_cast<const f32x4 *>(ldsf + L::wxf + 16 * t + 4 * g);
  constexpr int DB = (EFFECT == 2) ? 2 : 4;
  const int nd = (EFFECT == 2) ? 2 : n_doses;
  const int n_calls = (nd + 3) >> 2, n_own = (EFFECT == 1) ? (n_calls & ~3) : 0;
  f32x4 nz = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int kb = 0; kb < n_calls; ++kb) {
    BGM_NO_HOIST();
    const bool own = kb < n_own;
    const int c4 = kb & ~3, p4 = kb & 3;
    if (sample_y && (!own || p4 == 0)) nz = box_muller4(philox4x32_10(rowid, it, (unsigned)(own ? c4 + g : kb), TAG_YNOISE, k0, k1));
    float xk[DB];
#pragma unroll
    for (int e = 0; e < DB; ++e) {
      const int k = own ? 4 * (c4 + e) + p4 : 4 * kb + e;
      xk[e] = (EFFECT == 2) ? (e == 0 ? 1.0f : 0.0f) : x_values[k < nd ? k : nd - 1];
    }
    // layer 2 (64 -> 32): A fragments of a K block shared by the DB doses
    f32x4 a2[DB][2];
#pragma unroll
    for (int e = 0; e < DB; ++e) bx_bias<2>(ldsf, L::bf2, g, a2[e]);
    const bx_lds_ptr wf2 = bx_frag_base(lds + L::wf2, lane * 16);
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      bx_x8 ah[2], al[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        ah[mt] = bx_ld8(wf2, mt * 4096 + T * 2048);
        al[mt] = bx_ld8(wf2, mt * 4096 + T * 2048 + 1024);
      }
#pragma unroll
      for (int e = 0; e < DB; ++e) {
        f32x4 u0, u1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          u0[r] = lrelu_s(fmaf(wx[2 * T][r], xk[e], base[2 * T][r]));
          u1[r] = lrelu_s(fmaf(wx[2 * T + 1][r], xk[e], base[2 * T + 1][r]));
        }
        bx_x8 bh, bl;
        bx_split8(u0, u1, bh, bl);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a2[e][mt] = BX_MFMA32(al[mt], bh, a2[e][mt]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a2[e][mt] = BX_MFMA32(ah[mt], bl, a2[e][mt]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a2[e][mt] = BX_MFMA32(ah[mt], bh, a2[e][mt]);
      }
    }
    float mu[DB], sr[DB];
#pragma unroll
    for (int e = 0; e < DB; ++e) {
      bx_lrelu<2>(a2[e]);
      f32x4 a3[1];
      bx_bias<1>(ldsf, L::bf3, g, a3);
      bx_dense_acc<2, 1>(lds + L::wf3, lane, a2[e], a3);
      bx_lrelu<1>(a3);
      f32x4 a4[1];
      bx_bias<1>(ldsf, L::bf4, g, a4);
      bx_dense_acc<1, 1>(lds + L::wf4, lane, a3, a4);
      mu[e] = a4[0][0];
      sr[e] = a4[0][1];
    }
    if constexpr (EFFECT == 1) {
      const int k = own ? 4 * (c4 + g) + p4 : 4 * kb + g;
      const float mu_m = pick_by_group(g, mu[0], mu[1], mu[2], mu[3]), sr_m = pick_by_group(g, sr[0], sr[1], sr[2], sr[3]);
      const float s2 = (m.sig2_y > 0.0f) ? m.sig2_y : softplus_f(sr_m) + BGM_EPS;
      const float sd_m = __builtin_sqrtf(s2);
      if constexpr (CACHE) cache[kb * 64 + lane] = make_float2(mu_m, sd_m);      // for causal_effects_cached (causal_kernels.h)
      const float noise = own ? nz[0] : pick_by_group(g, nz[0], nz[1], nz[2], nz[3]);
      if (own) nz = f32x4{nz[1], nz[2], nz[3], nz[0]};
      float y = sample_y ? fmaf(sd_m, noise, mu_m) : mu_m;
      y = (valid && k < nd) ? y : 0.0f;
      const float tot = sum_over_j_to_lane15(y);
      if (j == 15 && k < nd) unsafeAtomicAdd(adrf_slot + (long long)d * nd + k, tot);
    } else {
      float yk[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float s2 = (m.sig2_y > 0.0f) ? m.sig2_y : softplus_f(sr[e]) + BGM_EPS;
        yk[e] = sample_y ? fmaf(__builtin_sqrtf(s2), nz[e], mu[e]) : mu[e];
      }
      if (g == 0 && row < n) ite[row * (long long)n_keep + d] = yk[0] - yk[1];
    }
  }
}

// ---------------------------------------------------------------------------
// Persistent random-walk Metropolis-Hastings over a segment of iterations (causal_mh_kernel in split precision; one row tile
// per wave, 8 waves per block: the two waves of a SIMD balance their progress through LDS as in the fp32 kernel).
// ---------------------------------------------------------------------------
template <int KT1, int NTL, int WAVES, int EFFECT>
__global__ __launch_bounds__(64 * WAVES) void causal_mh_bx3_kernel(CausalBxKArgs ka) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bx_lds[];
  const CausalMhKArgs &a = ka.a;
  const BxMeta &m = ka.bx;
  using L = BxLayout<KT1, NTL>;
  bx_lds_fill(bx_lds, ka.bblob, m.total_bytes);
  const float *ldsf = reinterpret_cast<const float *>(bx_lds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const long long n = a.n;
  const long long n_tiles = (n + 15) / 16;
  const long long slot = (long long)blockIdx.x * WAVES + wave;
  const long long n_slots = (long long)gridDim.x * WAVES;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // [WAVES] progress counters after the blob (LDS-typed: see causal_mh_kernel)
  volatile __attribute__((address_space(3))) int *prog =
      (volatile __attribute__((address_space(3))) int *)((__attribute__((address_space(3))) unsigned char *)bx_lds + m.total_bytes);
  if (lane == 0) prog[wave_u] = 0;
  const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();
  int tiles_done = 0;
  [[maybe_unused]] int ev_cnt = 0;          // EFFECT == 3 (event form of the retained phase, causal_event_kernels.h)
  for (long long tile = slot; tile < n_tiles; tile += n_slots) {
    const long long row0 = tile * 16;
    long long row = row0 + j;
    const bool valid = row < n;
    const long long rowc = valid ? row : n - 1;
    const float xr[1] = {a.x[rowc]}, yr = a.y[rowc];
    const unsigned rowid = (unsigned)(a.row_base + rowc);
    f32x4 vreg[NTL];
    bx_load_v<NTL>(a.v, ldsf + L::bgl, n, m.p, row0, j, g, vreg);
    f32x4 zs[1][KT1];
    float lp;
    // conditional latent prior (a.seg != NULL, wave-uniform): the row's (mu, 1 / sigma^2, (q / 2) log sigma^2) in registers, the difference to
    // the standard-normal term added to every log posterior in fp32 (PriorRow of causal_kernels.h, as its PRIOR = 1 instantiations do)
    const bool has_prior = a.seg != nullptr;
    PriorRow<KT1> pr;
    if (has_prior) pr.load(a.seg, a.prior_tab, rowc, m.q, g);
    if (a.init) {
#pragma unroll
      for (int t = 0; t < KT1; ++t) {
        const f32x4 e = box_muller4(philox4x32_10(rowid, 0u, (unsigned)(g + 4 * t), TAG_INIT, a.k0, a.k1));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * r + g;
          zs[0][t][r] = (f < m.q) ? e[r] : (f == m.q ? xr[0] : 0.0f);
        }
      }
      lp = causal_logp_bx3<KT1, NTL>(bx_lds, m, lane, g, j, zs[0], vreg, xr[0], yr);
      if (has_prior) lp += pr.correction(zs[0], m.q, g);
    } else {
      load_z_rows<KT1, 1>(a.state, n, m.q, row0, j, g, xr, zs);
      lp = a.logp[rowc];
    }
    uint4 uacc = make_uint4(0u, 0u, 0u, 0u);
    [[maybe_unused]] const int ev_tile0 = ev_cnt;
    bool eff_cached = false;      // outcome-net cache of the retained iterations: see causal_mh_kernel
    unsigned n_eff_skipped = 0u;
    for (int it = a.it_begin; it < a.it_begin + a.n_iters; ++it) {
      BGM_NO_HOIST();
      if constexpr (WAVES == 8) {
        const int mine = tiles_done * a.n_iters + (it - a.it_begin);
        if (lane == 0) prog[wave_u] = mine;
        const int other = __builtin_amdgcn_readfirstlane(prog[wave_u ^ 4]);
        if (mine < other) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
      }
      f32x4 zp[KT1];
#pragma unroll
      for (int t = 0; t < KT1; ++t) {
        const f32x4 e = box_muller4(philox4x32_10(rowid, (unsigned)it, (unsigned)(g + 4 * t), TAG_PROP, a.k0, a.k1));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * r + g;
          zp[t][r] = (f < m.q) ? fmaf(a.q_sd, e[r], zs[0][t][r]) : zs[0][t][r];
        }
      }
      float lpp = causal_logp_bx3<KT1, NTL>(bx_lds, m, lane, g, j, zp, vreg, xr[0], yr);
      if (has_prior) lpp += pr.correction(zp, m.q, g);
      if ((it & 3) == 0 || it == a.it_begin) uacc = philox4x32_10(rowid, (unsigned)it >> 2, 0u, TAG_ACC, a.k0, a.k1);
      const unsigned w = (it & 2) ? ((it & 1) ? uacc.w : uacc.z) : ((it & 1) ? uacc.y : uacc.x);
      const float u = u01_open(w);
      const float ratio = fast_exp(fminf(lpp - lp, 0.0f));
      const bool acc = u < ratio;
#pragma unroll
      for (int t = 0; t < KT1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) zs[0][t][r] = acc ? zp[t][r] : zs[0][t][r];
      lp = acc ? lpp : lp;
      const unsigned accmask = (unsigned)__popcll(__ballot(acc && valid && g == 0));
      if (a.acc_count != nullptr && lane == 0) atomicAdd(&a.acc_count[slot * (long long)a.n_iters + (it - a.it_begin)], accmask);
      if (it >= a.burn_in) {
        const long long d = it - a.burn_in;
        if (a.draws != nullptr && valid) {
          float *dr = a.draws + (d * n + row) * (long long)m.q;
#pragma unroll
          for (int t = 0; t < KT1; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int f = 16 * t + 4 * r + g;
              if (f < m.q) dr[f] = zs[0][t][r];
            }
        }
        if constexpr (EFFECT == 3) {
          // event form: the accepted moves (every chain at a call's first retained iteration) are appended to the slot's region; the
          // outcome net runs on the dense event tiles afterwards, in fp32 (causal_event_f_kernel on the fp32 sampling blob)
          causal_event_append<KT1>(a, m.q, slot, ev_cnt, valid && (acc || (a.ev_first && it == a.it_begin)), g, j, zs[0], it);
        } else if constexpr (EFFECT == 1) {
          float2 *cache = reinterpret_cast<float2 *>(a.eff_cache) + slot * (long long)((a.n_doses + 3) >> 2) * 64;
          float *adrf_slot = a.adrf_partial + slot * (long long)a.n_doses * a.n_keep;
          const bool skip = a.eff_skip && eff_cached && accmask == 0u;                   // wave-uniform: nobody moved
          causal_effects_bx3<KT1, NTL, EFFECT, true>(bx_lds, m, lane, g, j, zs[0], rowid, valid, row, n, (unsigned)it, d, a.n_keep, a.sample_y,
                                                     skip ? 0 : a.n_doses, a.x_values, adrf_slot, a.ite, a.k0, a.k1, cache);
          if (skip) causal_effects_cached(g, j, lane, rowid, valid, (unsigned)it, d, a.sample_y, a.n_doses, adrf_slot, a.k0, a.k1, cache);
          eff_cached = true;
          n_eff_skipped += skip ? 1u : 0u;
        } else if constexpr (EFFECT != 0) {
          causal_effects_bx3<KT1, NTL, EFFECT>(bx_lds, m, lane, g, j, zs[0], rowid, valid, row, n, (unsigned)it, d, a.n_keep, a.sample_y,
                                          a.n_doses, a.x_values,
                                          a.adrf_partial + slot * (long long)((EFFECT == 2) ? 2 : a.n_doses) * a.n_keep, a.ite, a.k0, a.k1);
        }
      }
    }
    if constexpr (EFFECT == 1) {
      if (a.eff_stats != nullptr && lane == 0 && n_eff_skipped != 0u) atomicAdd(&a.eff_stats[0], (unsigned long long)n_eff_skipped);
    }
    if constexpr (EFFECT == 3) {
      if (lane == 0) { a.tile_ev[2 * tile] = ev_tile0; a.tile_ev[2 * tile + 1] = ev_cnt - ev_tile0; }
    }
    store_z_rows<KT1, 1>(a.state, n, m.q, row0, j, g, zs);
    if (g == 0 && valid) a.logp[row] = lp;
    ++tiles_done;
  }
  if (lane == 0) prog[wave_u] = 0x7fffffff;
  if constexpr (EFFECT == 3) {
    if (lane == 0) a.slot_cnt[slot] = ev_cnt;
  }
  if (a.clk != nullptr && lane == 0) {   // [n_slots][4]: shader cycles, 100 MHz ticks, start tick, XCC id (as causal_mh_kernel)
    a.clk[4 * slot + 0] = __builtin_readcyclecounter() - clk_c0;
    a.clk[4 * slot + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    a.clk[4 * slot + 2] = clk_r0;
    a.clk[4 * slot + 3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
  }
}

#undef BX_MFMA32
#undef BX_MFMA16
}  // namespace BX_NS
