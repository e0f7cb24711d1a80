// bnf_kernels.h -- CausalBGM with Bayesian networks, input normalisation in inference mode (params['bnn_norm'] = "fixed", the
// shipped default): posterior sampling and causal effects on gfx950.
//
// replaces (src/bayesgm/models/causalbgm/base.py, use_bnn branches; networks/bnn.py:4-38):
//   get_log_posterior :765-817 + metropolis_hastings_sampler :820-904  -> bnf_noise_kernel, bnf_signs_kernel, bnf_mh_kernel
//   infer_from_latent_posterior :671-763                                -> bnf_effects_kernel
//
// With the input BatchNormalization on its (never updated) moving averages the rows of a block are independent: what a block of
// `bs` rows still shares is ONE weight perturbation dW = sigma * eps per (layer, call).  The batch-statistics kernels of
// bnn_sample_kernels.h (layer-synchronous staging of loc AND dW through 32 KB of LDS, three barriers per layer) stay for
// bnn_norm = "batch"; this file is built around the independence:
//   * persistent workgroups, one per CU; the `loc` fragments of g | h | f (150 KB at p = 200), the biases and the per-column
//     normalisation live in LDS for the whole launch (one fill per launch, one barrier in the kernel);
//   * a wave owns R 16-row tiles of ONE block at a time and walks g, h, f for the proposal and for the current state with the
//     activations in registers (swapped MFMA orientation: a layer's accumulators are the next layer's B operands);
//   * the call's perturbation fragments are read by every wave straight from L2 into registers, one output tile ahead of
//     their use (1 KB per 8 R MFMAs): no staging, no barriers; the workgroups of an XCD walk the blocks in the same order so
//     that a block's 150 KB set is fetched from HBM once per XCD;
//   * Rademacher sign words and perturbations are produced by their own launches (bnf_signs_kernel, bnf_noise_kernel): a
//     Philox call costs ~110 VALU instructions, 26 of them per row and iteration would sit in the matrix pipe's shadow
//     otherwise; the words come back as 16-byte loads one layer ahead;
//   * every layer extent is a compile-time constant (the default nets: g [q, 64 x 5, p + 1], f / h [., 64, 32, 8, 2]); a sign
//     flip is shift + v_and_or (-> +-1.0f) + one fma; LeakyReLU is one fma (lrelu_s, factor 0.6 folded into the packed weights).
// Fragment layouts (bnf_api.hip builds the index tables): the first layer of each net contracts over the SHARED extended input
// e = [z (q), x]: slot (lane group gg, register r) of k-tile sb holds e[16 sb + 4 r + gg] -- the layout one Philox call per lane
// fills (oracle/rng.py) -- with zero rows where a net does not read a component (h: z1, z3 and x; g: x; f: z2, z3); hidden layers
// are natural (feature 16 t + 4 gg + r); the 8 outputs of f / h's third layer sit at lane group f >> 1, register f & 1, so the
// last layer contracts over two k-steps; its two output columns are replicated over the four lane groups.
#pragma once
#include <hip/hip_runtime.h>

#include "bnn_kernels.h"

#define BNF_NG_G 6            // 16-byte sign groups of one g call per row: G0 = [w0 w1 w2 -], Gl = [w(4l-1) .. w(4l+2)] (l = 1..4), G5 = [w19 w20 - -]
#define BNF_NG_H 3            // ... of one h / f call: H0 = [w0 w1 w2 -], H1 = [w3 w4 w5 w6], H2 = [w7 w8 w9 -]
#define BNF_GOUT 8            // sign words of the outputs of g's last layer, w21 .. w27 (p + 1 <= 208 columns), one spare
#define BNF_MAX_DOSES 256
// hipcc sinks every load to just above its first use; a compiler-only memory fence behind a group of prefetch loads keeps them
// where they were requested (one output tile / one layer / one dose ahead of their use)
#define BNF_PIN() asm volatile("" ::: "memory")
// -D BNF_PROF: cycle stamps of wave 0 of every workgroup per section (0 prologue, 1 g first, 2 g hidden, 3 g last, 4 h, 5 f, 6 accept / rest),
// summed into BnfMhArgs::prof
#ifdef BNF_PROF
#define BNF_T(k) { const unsigned long long t__ = __builtin_readcyclecounter(); bnf_tp[k] += t__ - bnf_tl; bnf_tl = t__; }
#define BNF_PROF_PARAM , unsigned long long (&bnf_tp)[8], unsigned long long &bnf_tl
#define BNF_PROF_ARG , bnf_tp, bnf_tl
#else
#define BNF_T(k)
#define BNF_PROF_PARAM
#define BNF_PROF_ARG
#endif

typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

struct BnfPlan {
  int q, p, z0, z1, z2, binary;
  int KS;                     // k-steps of the first layers: ceil((q + 1) / 4); k-tiles T0 = (KS + 3) / 4
  int NTL;                    // output tiles of g's last layer: ceil((p + 1) / 16)
  int fg0, fgh, fgl, fh, ff;  // first fragment (256 floats) of g's first / hidden / last layers, of h, of f
  int n_frags;
  int bias_off, norm_off, shift_off, blob_floats;   // float offsets inside the LDS blob [frags | bias tiles | norm | shift]
  int bg0, bgh, bgl, bh, bf;  // bias tiles (16 floats) of the same
  int set_floats;             // one perturbation set: n_frags * 256
  int lds_skip;               // floats of the blob that stay out of the LDS (WIDE: g's last layer, placed last among the fragments), else 0
  // effects blob (outcome net only): [f frags | f bias tiles | f norm | f shift].  Its first layer contracts over the net's OWN input
  // (z0, z1, x) -- slot (gg, r) of k-tile sb = input 16 sb + 4 r + gg -- in KSF = ceil((z0 + z1 + 1) / 4) k-steps (1 for z_dims
  // [1, 1, 1, 7] against the 3 of the shared extended input)
  int KSF, e_frags, e_bias_off, e_norm_off, e_shift_off, e_blob_floats;
};

struct BnfSigns {             // n_states = 2 (MH: proposal, current) or n_doses (effects)
  const uint4 *g;             // [n_states][BNF_NG_G][n]
  const uint32_t *gout;       // [n_states][n][BNF_GOUT]
  const uint4 *h, *f;         // [n_states][BNF_NG_H][n]
};

// softplus_f (bgm_device.h) without control flow: both branches of its small-argument split are evaluated and selected (hipcc
// turns the plain ternary into an exec-mask branch around the v_log_f32, which splits the scheduling regions of the callers)
__device__ __forceinline__ float bnf_softplus(float x) {
  const float e = fast_exp(-fabsf(x));
  float l = fast_log(1.0f + e), sm = e * (1.0f - 0.5f * e);
  asm volatile("" : "+v"(l), "+v"(sm));
  return vmax(x, 0.0f) + ((e < 2.44140625e-4f) ? sm : l);
}
// +-1.0f from the bit at position 31 of m
// The mask comes out of an (opaque, CSE-able) asm so that it lives in an SGPR: with the literal 0x80000000 the compiler emits
// v_and_b32 + v_or_b32 (VOP3 takes no literal on gfx9), with a register operand one v_and_or_b32.
__device__ __forceinline__ uint32_t bnf_k80() { uint32_t k; asm("s_mov_b32 %0, 0x80000000" : "=s"(k)); return k; }
#ifdef BNF_ABL_NOSIGN      // development ablation (never in the product build): every Rademacher sign = +1, i.e. the sign applications
// cost nothing at all -- the upper bound of what ANY cheaper form of them (lane masks, packed flips) could gain
__device__ __forceinline__ float bnf_pm1(uint32_t) { return 1.0f; }
#else
__device__ __forceinline__ float bnf_pm1(uint32_t m) { return __builtin_bit_cast(float, (m & bnf_k80()) | 0x3f800000u); }
#endif
// sign words are pre-shifted per lane so that bit (4 gg + 19) of the word sits at position 31: feature 16 t + 4 gg + r of a
// 32-feature word <-> bit B = 16 (t & 1) + r of the lane's view, at position 12 + B
__device__ __forceinline__ uint32_t bnf_preshift(uint32_t w, int g) { return w << (12 - 4 * g); }
template <int B>
__device__ __forceinline__ float bnf_sign(uint32_t wsh) { return bnf_pm1(B == 19 ? wsh : wsh << (19 - B)); }
__device__ __forceinline__ float bnf_sign_rt(uint32_t wsh, int b) { return bnf_pm1(wsh << (19 - b)); }

struct BnfLds {
  const f32x4 *frag;          // fragment f at frag + 64 f (one f32x4 per lane)
  const f32x4 *bias;          // tile t, lane group gg at bias[4 t + gg]
  const f32x4 *norm;          // net N (0 g, 1 h, 2 f), k-tile sb: scale at norm[((N * T0 + sb) * 2) * 4 + gg], shift at [... + 1) * 4 + gg]
  const int4 *shift;          // net N, k-tile sb: left shifts that bring the slot's input-sign bit to position 31, [(N * T0 + sb) * 4 + gg]
};

// ---------------------------------------------------------------------------------------------
// first layer: extended input e (KS slots per lane) -> 64 units.  ze: raw slots; the layer normalises (scale / shift per slot and
// net), flips with the input-sign word w_in, runs both products and finishes the four output tiles.
// wo: output-sign words (pre-shifted), wi: input-sign words of the NEXT layer (pre-shifted).
// ---------------------------------------------------------------------------------------------
// DET: deterministic nets (BaseFullyConnectedNet, networks/base.py:4-51) = the same walk without the perturbation product, the sign
// flips and their loads: y = loc^T h + b
template <int KS, int R, bool DET = false>
__device__ __forceinline__ void bnf_first(const f32x4 *LF, const f32x4 *__restrict__ DW, const f32x4 *BL, const f32x4 *NORM, const int4 *SHIFT,
                                          int lane, int g, const float (&ze)[R][KS], const uint32_t (&w_in)[R], const uint32_t (&wo)[R][2],
                                          const uint4 (&Gn)[R], float (&h)[R][4][4], float (&hs)[R][4][4]) {
  constexpr int T0 = (KS + 3) / 4;
  f32x4 fd[4][T0];
  if constexpr (!DET) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int sb = 0; sb < T0; ++sb) fd[mt][sb] = DW[(mt * T0 + sb) * 64 + lane];
    BNF_PIN();
  }
  float hb[R][KS], hsb[R][KS];
#pragma unroll
  for (int sb = 0; sb < T0; ++sb) {
    const f32x4 sc = NORM[(sb * 2) * 4 + g], sh = NORM[(sb * 2 + 1) * 4 + g];
    const int4 st = SHIFT[sb * 4 + g];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ks = 4 * sb + r;
      if (ks < KS) {
        const int sft = r == 0 ? st.x : r == 1 ? st.y : r == 2 ? st.z : st.w;
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
          const float x = fmaf(ze[rt][ks], sc[r], sh[r]);
          hb[rt][ks] = x;
          if constexpr (!DET) hsb[rt][ks] = x * bnf_pm1(w_in[rt] << sft);
        }
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    f32x4 a1[R], a2[R];
    const f32x4 b = BL[4 * mt + g];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { a1[rt] = b; a2[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int sb = 0; sb < T0; ++sb) {
      const f32x4 fa = LF[(mt * T0 + sb) * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * sb + r < KS) {
#pragma unroll
          for (int rt = 0; rt < R; ++rt) {
            a1[rt] = BGM_MFMA(fa[r], hb[rt][4 * sb + r], a1[rt]);
            if constexpr (!DET) a2[rt] = BGM_MFMA(fd[mt][sb][r], hsb[rt][4 * sb + r], a2[rt]);
          }
        }
    }
#pragma unroll
    for (int rt = 0; rt < R; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (DET) h[rt][mt][r] = lrelu_s(a1[rt][r]);
        else {
          const float y = fmaf(a2[rt][r], bnf_sign_rt(wo[rt][mt >> 1], 16 * (mt & 1) + r), a1[rt][r]);
          const float v = lrelu_s(y);
          h[rt][mt][r] = v;
          hs[rt][mt][r] = v * bnf_sign_rt(bnf_preshift(mt < 2 ? Gn[rt].x : Gn[rt].y, g), 16 * (mt & 1) + r);
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
// hidden layer 64 -> 64 of g: h, hs are replaced by the next layer's inputs.  fd: this layer's first tile's perturbation
// fragments (requested by the caller); on return the fragments at DWnext (the next layer's / section's first tile).
// ---------------------------------------------------------------------------------------------
// development ablations (never defined in the product build): BNF_ABL_NODW reads the perturbation fragments from LDS (the loc
// fragments) instead of L2; BNF_ABL_NOEPI replaces the sign / activation epilogues of g's hidden and last layers by a copy
#ifdef BNF_ABL_NODW
#define BNF_DWSRC(dw, lf) (lf)
#else
#define BNF_DWSRC(dw, lf) (dw)
#endif
template <int R, bool DET = false>
__device__ __forceinline__ void bnf_hidden(const f32x4 *LF, const f32x4 *__restrict__ DW, const f32x4 *__restrict__ DWnext, const f32x4 *BL, int lane,
                                           int g, const uint32_t (&wo)[R][2], const uint4 (&Gn)[R], const float (&h)[R][4][4],
                                           const float (&hs)[R][4][4], float (&hn)[R][4][4], float (&hsn)[R][4][4], f32x4 (&fd)[4]) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    f32x4 fn[4];
    if constexpr (!DET) {
      const f32x4 *nx = (mt < 3) ? BNF_DWSRC(DW, LF) + (mt + 1) * 4 * 64 : BNF_DWSRC(DWnext, LF);
#pragma unroll
      for (int t = 0; t < 4; ++t) fn[t] = nx[t * 64 + lane];
      BNF_PIN();
    } else BGM_NO_HOIST();
    f32x4 a1[R], a2[R];
    const f32x4 b = BL[4 * mt + g];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { a1[rt] = b; a2[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 fa = LF[(mt * 4 + t) * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
          a1[rt] = BGM_MFMA(fa[r], h[rt][t][r], a1[rt]);
          if constexpr (!DET) a2[rt] = BGM_MFMA(fd[t][r], hs[rt][t][r], a2[rt]);
        }
    }
#pragma unroll
    for (int rt = 0; rt < R; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (DET) { hn[rt][mt][r] = lrelu_s(a1[rt][r]); continue; }
#ifdef BNF_ABL_NOEPI
        hn[rt][mt][r] = a1[rt][r]; hsn[rt][mt][r] = a2[rt][r];
#else
        const float y = fmaf(a2[rt][r], bnf_sign_rt(wo[rt][mt >> 1], 16 * (mt & 1) + r), a1[rt][r]);
        const float v = lrelu_s(y);
        hn[rt][mt][r] = v;
        hsn[rt][mt][r] = v * bnf_sign_rt(bnf_preshift(mt < 2 ? Gn[rt].x : Gn[rt].y, g), 16 * (mt & 1) + r);      // Gn: requested at the layer's start, first touched here
#endif
      }
    if constexpr (!DET) {
#pragma unroll
      for (int t = 0; t < 4; ++t) fd[t] = fn[t];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// outcome / treatment net  e -> 64 -> 32 -> 8 -> 2  (f, h of the default configuration).  G: the call's three sign groups per
// row tile.  Returns the two outputs (mean, raw variance) of every row, valid in all four lane groups.
// LF / DW: the net's first fragment in LDS / in the call's perturbation set; BL: its first bias tile.
// ---------------------------------------------------------------------------------------------
template <int KS, int R, bool DET = false>
__device__ __forceinline__ void bnf_head(const f32x4 *LF, const f32x4 *__restrict__ DW, const f32x4 *BL, const f32x4 *NORM, const int4 *SHIFT, int lane,
                                         int g, const float (&ze)[R][KS], const uint4 (&G)[R][BNF_NG_H], float (&mu)[R], float (&raw)[R]) {
  constexpr int T0 = (KS + 3) / 4;
  // perturbation fragments of the small layers 2..4 are requested one layer ahead (they have no tile loop to hide them in): layer 2's
  // first output tile before layer 1, its second tile and layers 3 / 4 behind layer 1 -- all eleven up front cost 44 registers at the
  // point where the two-tile sampler is fullest (43 spilled)
  f32x4 fd2[2][4], fd3[2], fd4;
  const f32x4 *D2 = DW + 4 * T0 * 64;
  if constexpr (!DET) {
#pragma unroll
    for (int t = 0; t < 4; ++t) fd2[0][t] = D2[t * 64 + lane];
    BNF_PIN();
  }
  float h1[R][4][4], hs1[R][4][4];
  {
    uint32_t w_in[R], wo[R][2];
    uint4 Gn[R];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      w_in[rt] = G[rt][0].x;
      wo[rt][0] = bnf_preshift(G[rt][0].y, g); wo[rt][1] = bnf_preshift(G[rt][0].z, g);
      Gn[rt] = G[rt][1];
    }
    bnf_first<KS, R, DET>(LF, DW, BL, NORM, SHIFT, lane, g, ze, w_in, wo, Gn, h1, hs1);
  }
  if constexpr (!DET) {
#pragma unroll
    for (int t = 0; t < 4; ++t) fd2[1][t] = D2[(4 + t) * 64 + lane];
    fd3[0] = D2[8 * 64 + lane]; fd3[1] = D2[9 * 64 + lane];
    fd4 = D2[10 * 64 + lane];
    BNF_PIN();
  }
  // layer 2: 64 -> 32
  const f32x4 *L2 = LF + 4 * T0 * 64;
  float h2[R][2][4], hs2[R][2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    f32x4 a1[R], a2[R];
    const f32x4 b = BL[4 * (4 + mt) + g];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { a1[rt] = b; a2[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 fa = L2[(mt * 4 + t) * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
          a1[rt] = BGM_MFMA(fa[r], h1[rt][t][r], a1[rt]);
          if constexpr (!DET) a2[rt] = BGM_MFMA(fd2[mt][t][r], hs1[rt][t][r], a2[rt]);
        }
    }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      if constexpr (DET) {
#pragma unroll
        for (int r = 0; r < 4; ++r) h2[rt][mt][r] = lrelu_s(a1[rt][r]);
      } else {
        const uint32_t so = bnf_preshift(G[rt][1].z, g), si = bnf_preshift(G[rt][1].w, g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y = fmaf(a2[rt][r], bnf_sign_rt(so, 16 * mt + r), a1[rt][r]);
          const float v = lrelu_s(y);
          h2[rt][mt][r] = v;
          hs2[rt][mt][r] = v * bnf_sign_rt(si, 16 * mt + r);
        }
      }
    }
  }
  // layer 3: 32 -> 8, output feature f at lane group f >> 1, register f & 1
  const f32x4 *L3 = L2 + 8 * 64;
  float h3[R][2], hs3[R][2];
  {
    f32x4 a1[R], a2[R];
    const f32x4 b = BL[4 * 6 + g];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { a1[rt] = b; a2[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4 fa = L3[t * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
          a1[rt] = BGM_MFMA(fa[r], h2[rt][t][r], a1[rt]);
          if constexpr (!DET) a2[rt] = BGM_MFMA(fd3[t][r], hs2[rt][t][r], a2[rt]);
        }
    }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      if constexpr (DET) {
        h3[rt][0] = lrelu_s(a1[rt][0]); h3[rt][1] = lrelu_s(a1[rt][1]);
      } else {
        const uint32_t so = G[rt][2].x << (30 - 2 * g), si = G[rt][2].y << (30 - 2 * g);   // bit 2 gg + r at position 30 + r
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const float y = fmaf(a2[rt][r], bnf_pm1(r ? so : so << 1), a1[rt][r]);
          const float v = lrelu_s(y);
          h3[rt][r] = v;
          hs3[rt][r] = v * bnf_pm1(r ? si : si << 1);
        }
      }
    }
  }
  // layer 4: 8 -> 2 over two k-steps; column o at every 4 gg + o
  {
    const f32x4 fa = L3[2 * 64 + lane];
    const f32x4 b = BL[4 * 7 + g];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      f32x4 a1 = b, a2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        a1 = BGM_MFMA(fa[r], h3[rt][r], a1);
        if constexpr (!DET) a2 = BGM_MFMA(fd4[r], hs3[rt][r], a2);
      }
      if constexpr (DET) { mu[rt] = a1[0]; raw[rt] = a1[1]; }
      else {
        const uint32_t so = G[rt][2].z;
        mu[rt] = fmaf(a2[0], bnf_pm1(so << 31), a1[0]);
        raw[rt] = fmaf(a2[1], bnf_pm1(so << 30), a1[1]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// log posterior of R row tiles for one state (one noisy call of g, h, f)
// ---------------------------------------------------------------------------------------------
struct BnfMhArgs {
  BnfPlan pl;
  const float *blob;                   // [blob_floats]
  const float *dw;                     // [n_blocks][n_states][set_floats]
  BnfSigns sg;
  const float *x, *y, *v;
  float *z;                            // [n x q] state (mode 1: updated in place; mode 0: read)
  const float *zprop;                  // MODE 3: [n x q] proposals (state slot 0; z is state slot 1)
  long long n, row_base;
  int bs, n_blocks, block0, groups_per_block, n_items, n_states;
  int mode;                            // 0: log posterior of z (state slot 0) -> out; 1: one MH iteration
  int it, init;
  float q_sd;
  const float *q_sd_blocks;
  uint32_t k0, k1;
  float *out;
  unsigned *acc_count, *acc_blocks;
  unsigned *queue;                     // [8] item counters, one per XCD (zeroed by bnf_signs_kernel / by the host for deterministic nets)
  float *lp_cache;                     // deterministic nets, MODE 1: [n] log posterior of the current states (in / out)
  double *sums;                        // MODE 2: [3] += sum |v - mu_v|^2, sum (x - x_pred)^2, sum (y - mu_y)^2
  float sig2_v, sig2_x, sig2_y;        // fixed variances params['sigma_*']^2 (<= 0: the networks' variance heads)
  // conditional latent prior (IdentifiableCausalBGM, bprior_kernels.h): [n_states][n][q + 2] = mu [q], 1 / sigma^2, (q / 2) log sigma^2 of
  // every row for the state's call of the prior net, or NULL = N(0, I)
  const float *prior;
  long long prior_stride;              // floats between the states' tables
  unsigned long long *prof;            // -D BNF_PROF
};

// The prior term of a row's log posterior through the standard-normal slot of bnf_logpost_rows: it takes zz, the lane's share of
// |z|^2, and returns  ... - zz / 2  summed over the lane groups; with a conditional prior the share becomes (z - mu)^2 / sigma^2 over the
// lane's features and (q / 2) log sigma^2 is subtracted afterwards (returned in lc).
template <int KS>
__device__ __forceinline__ float bnf_prior_share(const float *pr, int q, int g, const float (&z)[KS], float &lc) {
  float s = 0.0f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int f = 16 * (ks >> 2) + 4 * (ks & 3) + g;
    const float d = z[ks] - pr[min(f, q - 1)];
    s = f < q ? fmaf(d, d, s) : s;
  }
  lc = pr[q + 1];
  return s * pr[q];
}

// DET: deterministic nets.  WIDE: g's last layer does not fit the LDS next to the rest (p > 207 at the default widths): its loc
// fragments stay in the packed blob in HBM / L2 and are requested one output tile ahead, like the perturbations (the workgroups of
// an XCD walk the same fragments at about the same time, so they come from L2).  EVAL: instead of the log posterior return the
// pieces CausalBGM.evaluate needs (base.py:534-570): aux = {sum_j (mu_v - v)^2, mu_x (logit for a binary model), mu_y}.
template <int KS, int R, bool DET = false, bool WIDE = false, bool EVAL = false>
__device__ __forceinline__ void bnf_logpost_rows(const BnfMhArgs &a, const BnfLds &L, int lane, int j, int g, long long blk_lo, const int (&rib)[R],
                                                 const float (&ze)[R][KS], const float (&xr)[R], const float (&yr)[R], const float *dwset, int s,
                                                 const float (&zz)[R], float (&lp)[R], float (&aux)[R][3] BNF_PROF_PARAM) {
  constexpr int T0 = (KS + 3) / 4;
  const BnfPlan &P = a.pl;
  const int p = P.p;
  const f32x4 *DW = (const f32x4 *)dwset;
  BGM_NO_HOIST();       // the LDS-resident fragments never change: without a fence LICM hoists their reads out of the item loop and spills
  BNF_T(0);
  // ---- g
  float ssq[R], rawv[R];
  uint4 GH[R][BNF_NG_H];       // sign groups of the h call: requested before g's last layer, first touched after it
  {
    float h[R][4][4], hs[R][4][4];
    // wave-uniform bases of the block's rows; per-lane offsets stay 32-bit
    const uint4 *SG = DET ? nullptr : a.sg.g + ((long long)s * BNF_NG_G * a.n + blk_lo);
    uint4 Gc[R], Gn[R];
    if constexpr (!DET) {
#pragma unroll
      for (int rt = 0; rt < R; ++rt) { Gc[rt] = SG[rib[rt]]; Gn[rt] = (SG + a.n)[rib[rt]]; }
      BNF_PIN();
    }
    uint32_t wo[R][2];
    {
      uint32_t w_in[R];
      if constexpr (!DET) {
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
          w_in[rt] = Gc[rt].x;
          wo[rt][0] = bnf_preshift(Gc[rt].y, g); wo[rt][1] = bnf_preshift(Gc[rt].z, g);
        }
      }
      bnf_first<KS, R, DET>(L.frag + P.fg0 * 64, DW + P.fg0 * 64, L.bias + 4 * P.bg0, L.norm, L.shift, lane, g, ze, w_in, wo, Gn, h, hs);
    }
    BNF_T(1);
    f32x4 fd[4];
    if constexpr (!DET) {
      const f32x4 *D = DW + P.fgh * 64;
#pragma unroll
      for (int t = 0; t < 4; ++t) fd[t] = D[t * 64 + lane];
    }
    // two layers per trip: h -> hb -> h, so that no activation is copied at the loop's back edge
#pragma nounroll
    for (int l = 1; l <= 4; l += 2) {
      float hb[R][4][4], hsb[R][4][4];
      // out-sign words of a layer: the second half of the group requested one layer ago; the next group is requested now and first
      // touched in this layer's first epilogue
      if constexpr (!DET) {
#pragma unroll
        for (int rt = 0; rt < R; ++rt) { wo[rt][0] = bnf_preshift(Gn[rt].z, g); wo[rt][1] = bnf_preshift(Gn[rt].w, g); }
#pragma unroll
        for (int rt = 0; rt < R; ++rt) Gn[rt] = (SG + (long long)(l + 1) * a.n)[rib[rt]];
        BNF_PIN();
      }
      int fo = (P.fgh + 16 * (l - 1)) * 64;
      bnf_hidden<R, DET>(L.frag + fo, DW + fo, DW + fo + 16 * 64, L.bias + 4 * (P.bgh + 4 * (l - 1)), lane, g, wo, Gn, h, hs, hb, hsb, fd);
      if constexpr (!DET) {
#pragma unroll
        for (int rt = 0; rt < R; ++rt) { wo[rt][0] = bnf_preshift(Gn[rt].z, g); wo[rt][1] = bnf_preshift(Gn[rt].w, g); }
#pragma unroll
        for (int rt = 0; rt < R; ++rt) Gn[rt] = (SG + (long long)(l + 2) * a.n)[rib[rt]];
        BNF_PIN();
      }
      fo += 16 * 64;
      // the fragments behind the last hidden layer are those of g's last layer, wherever the plan put them
      bnf_hidden<R, DET>(L.frag + fo, DW + fo, (l + 1 < 4) ? DW + fo + 16 * 64 : DW + P.fgl * 64, L.bias + 4 * (P.bgh + 4 * l), lane, g, wo, Gn, hb,
                         hsb, h, hs, fd);
    }
    BNF_T(2);
    if constexpr (!DET) {
      const uint4 *SH = a.sg.h + ((long long)s * BNF_NG_H * a.n + blk_lo);
#pragma unroll
      for (int rt = 0; rt < R; ++rt)
#pragma unroll
        for (int k = 0; k < BNF_NG_H; ++k) GH[rt][k] = (SH + (long long)k * a.n)[rib[rt]];
    }
    // last layer: 64 -> p + 1, tile by tile against the data row; fd holds tile 0's perturbation fragments
    const int NTL = P.NTL;
    const f32x4 *LFl = L.frag + P.fgl * 64, *DWl = DW + P.fgl * 64, *BLl = L.bias + 4 * P.bgl;
    const f32x4 *LFg = (const f32x4 *)a.blob + P.fgl * 64;          // WIDE: the same fragments in the packed blob (HBM / L2)
    const uint32_t *GO = DET ? nullptr : a.sg.gout + ((long long)s * a.n + blk_lo) * BNF_GOUT;
    const float *vblk = a.v + blk_lo * p;
    // the lane's four columns of the data row as ONE 16-byte request (dword-aligned: p need not be a multiple of 4); nothing is
    // done with the value before the tile that consumes it, so the request stays in flight for a whole tile.  Only the last tile
    // has columns >= p: it clamps the address and masks at its use.
    auto load_v = [&](int rt, int mt) __attribute__((always_inline)) -> f32x4 {
      const float *vr = vblk + rib[rt] * p + min(16 * mt + 4 * g, p - 4);
      return *(const f32x4_u *)vr;
    };
    f32x4 vn[R], fw[4];
    uint32_t wn[R];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { vn[rt] = load_v(rt, 0); wn[rt] = DET ? 0u : GO[rib[rt] * BNF_GOUT]; ssq[rt] = 0.0f; rawv[rt] = 0.0f; }
    if constexpr (WIDE) {
#pragma unroll
      for (int t = 0; t < 4; ++t) fw[t] = LFg[t * 64 + lane];
    }
    auto tile = [&](int mt, bool last, const f32x4 (&vc)[R], const uint32_t (&wc)[R], const f32x4 (&fc)[4], const f32x4 (&fwc)[4]) __attribute__((always_inline)) {
      f32x4 a1[R], a2[R];
      const f32x4 b = BLl[4 * mt + g];
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        a1[rt] = b - vc[rt];
        if (last) {      // columns u >= p (the variance column and the padding) have no data; a clamped request is shifted back
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
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 fa;
        if constexpr (WIDE) fa = fwc[t]; else fa = LFl[(mt * 4 + t) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int rt = 0; rt < R; ++rt) {
            a1[rt] = BGM_MFMA(fa[r], h[rt][t][r], a1[rt]);
            if constexpr (!DET) a2[rt] = BGM_MFMA(fc[t][r], hs[rt][t][r], a2[rt]);
          }
      }
      const int pos = 16 * (mt & 1);
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        const uint32_t wsh = bnf_preshift(wc[rt], g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float d;
          if constexpr (DET) d = a1[rt][r];
          else {
#ifdef BNF_ABL_NOEPI
            d = a1[rt][r] + a2[rt][r];
#else
            d = fmaf(a2[rt][r], bnf_sign_rt(wsh, pos + r), a1[rt][r]);
#endif
          }
          if (!last) ssq[rt] = fmaf(d, d, ssq[rt]);
          else {
            const int u = 16 * mt + 4 * g + r;
            ssq[rt] = fmaf(u < p ? d : 0.0f, d, ssq[rt]);
            rawv[rt] += (u == p) ? d : 0.0f;
          }
        }
      }
    };
#pragma nounroll
    for (int mt = 0; mt < NTL - 1; ++mt) {
      f32x4 vc[R], fc[4], fwc[4];
      uint32_t wc[R];
#pragma unroll
      for (int rt = 0; rt < R; ++rt) { vc[rt] = vn[rt]; wc[rt] = wn[rt]; }
#pragma unroll
      for (int t = 0; t < 4; ++t) { fc[t] = fd[t]; fwc[t] = fw[t]; }
#pragma unroll
      for (int rt = 0; rt < R; ++rt) { vn[rt] = load_v(rt, mt + 1); if constexpr (!DET) wn[rt] = GO[rib[rt] * BNF_GOUT + ((mt + 1) >> 1)]; }
      if constexpr (!DET) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fd[t] = BNF_DWSRC(DWl, LFl)[((mt + 1) * 4 + t) * 64 + lane];
      }
      if constexpr (WIDE) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fw[t] = LFg[((mt + 1) * 4 + t) * 64 + lane];
      }
      BNF_PIN();
      tile(mt, false, vc, wc, fc, fwc);
    }
    tile(NTL - 1, true, vn, wn, fd, fw);
    BNF_T(3);
  }
  float part[R];
#pragma unroll
  for (int rt = 0; rt < R; ++rt) {
    if constexpr (EVAL) { aux[rt][0] = sum_over_g(ssq[rt]); part[rt] = 0.0f; lp[rt] = 0.0f; }
    else {
      const float rw = sum_over_g(rawv[rt]);
      const float s2 = (a.sig2_v > 0.0f) ? a.sig2_v : bnf_softplus(rw) + BGM_EPS;
      // the per-lane share of -(ssq / (2 s2) + |z|^2 / 2); the log term is added once after the cross-lane sum
      part[rt] = -(ssq[rt] * fast_rcp(2.0f * s2) + 0.5f * zz[rt]);
      lp[rt] = -0.5f * (float)p * fast_log(s2);
    }
  }
  // ---- h (treatment) and f (outcome)
  {
    float mu[R], raw[R];
    uint4 G[R][BNF_NG_H];
    if constexpr (!DET) {     // the f call's groups are requested before the h call and first touched after it
      const uint4 *SF = a.sg.f + ((long long)s * BNF_NG_H * a.n + blk_lo);
#pragma unroll
      for (int rt = 0; rt < R; ++rt)
#pragma unroll
        for (int k = 0; k < BNF_NG_H; ++k) G[rt][k] = (SF + (long long)k * a.n)[rib[rt]];
    }
    BGM_NO_HOIST();
    bnf_head<KS, R, DET>(L.frag + P.fh * 64, DW + P.fh * 64, L.bias + 4 * P.bh, L.norm + 1 * T0 * 8, L.shift + 1 * T0 * 4, lane, g, ze, GH, mu, raw);
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      const float m_ = mu[rt];
      if constexpr (EVAL) aux[rt][1] = m_;
      else if (P.binary) lp[rt] -= fmaxf(m_, 0.0f) - m_ * xr[rt] + bnf_softplus(-fabsf(m_));
      else {
        const float s2 = (a.sig2_x > 0.0f) ? a.sig2_x : bnf_softplus(raw[rt]) + BGM_EPS, d = xr[rt] - m_;
        lp[rt] -= d * d * fast_rcp(2.0f * s2) + 0.5f * fast_log(s2);
      }
    }
    BNF_T(4);
    BGM_NO_HOIST();
    bnf_head<KS, R, DET>(L.frag + P.ff * 64, DW + P.ff * 64, L.bias + 4 * P.bf, L.norm + 2 * T0 * 8, L.shift + 2 * T0 * 4, lane, g, ze, G, mu, raw);
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      if constexpr (EVAL) aux[rt][2] = mu[rt];
      else {
        const float s2 = (a.sig2_y > 0.0f) ? a.sig2_y : bnf_softplus(raw[rt]) + BGM_EPS, d = yr[rt] - mu[rt];
        lp[rt] -= d * d * fast_rcp(2.0f * s2) + 0.5f * fast_log(s2);
      }
    }
  }
  if constexpr (!EVAL) {
#pragma unroll
    for (int rt = 0; rt < R; ++rt) lp[rt] += sum_over_g(part[rt]);
  }
  BNF_T(5);
}

// split-precision ("f16 x 3") forms of the network walks, defined in bnx_kernels.h (included behind this file by the translation unit that
// instantiates the X3 kernels): same arguments, fragment indices and sign-word protocol
template <int KS, int R>
__device__ __forceinline__ void bnx_logpost_rows(const BnfMhArgs &a, const BnfLds &L, int lane, int j, int g, long long blk_lo, const int (&rib)[R],
                                                 const float (&ze)[R][KS], const float (&xr)[R], const float (&yr)[R], const float *dwset, int s,
                                                 const float (&zz)[R], float (&lp)[R] BNF_PROF_PARAM);
template <int KS, int R>
__device__ __forceinline__ void bnx_head(const f32x4 *LF, const f32x4 *__restrict__ DW, const f32x4 *BL, const f32x4 *NORM, const int4 *SHIFT, int lane,
                                         int g, const float (&ze)[R][KS], const uint4 (&G)[R][BNF_NG_H], float (&mu)[R], float (&raw)[R]);

// ---------------------------------------------------------------------------------------------
// persistent sampler kernel: grid = a multiple of 8 workgroups (one per CU), WAVES waves each.  Items = groups of R row tiles of
// one block; the workgroups of XCD x (blockIdx % 8) take the x-th contiguous eighth of the items and walk it in lock step.
// ---------------------------------------------------------------------------------------------
// MODE 0: log posterior of the given states -> out.  MODE 3: the two log posteriors of a Metropolis-Hastings iteration (proposals zprop
// with perturbation set / sign state 0, current states z with set / state 1) -> out [2][n]; proposal and accept step are
// bnx_propose_kernel / bnx_accept_kernel's.  MODE 1: one Metropolis-Hastings iteration; with Bayesian nets both states are
// evaluated afresh, with deterministic nets (DET) the current state's value is carried in lp_cache like the reference's
// deterministic result would be (oracle/causal.py mh_transition).  MODE 2 (DET): the sums of CausalBGM.evaluate.
// X3: the networks in split precision (bnx_kernels.h; a.blob / a.dw in its fragment encoding)
template <int KS, int R, int WAVES, int MODE, bool DET = false, bool WIDE = false, bool X3 = false>
static __global__ __launch_bounds__(64 * WAVES, WAVES / 4) void bnf_mh_kernel(BnfMhArgs a) {
  static_assert(!X3 || (!DET && !WIDE && MODE != 2), "bnf_mh_kernel: split precision serves the Bayesian nets' log posterior and MH modes");
  static_assert(MODE != 3 || !DET, "bnf_mh_kernel: MODE 3 evaluates both states afresh (Bayesian nets)");
  extern __shared__ __attribute__((aligned(16))) float bnf_lds[];
  const BnfPlan &P = a.pl;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {   // LDS <- blob; a WIDE plan keeps g's last layer (the last fragments of the blob) out: lds_skip floats before the bias tiles
    const f32x4 *src = (const f32x4 *)a.blob;
    f32x4 *dst = (f32x4 *)bnf_lds;
    const int head = (WIDE ? P.fgl * 256 : P.blob_floats) >> 2, cnt = (P.blob_floats - P.lds_skip) >> 2, skip = P.lds_skip >> 2;
    for (int i = tid; i < cnt; i += 64 * WAVES) dst[i] = src[i < head ? i : i + skip];
  }
  __syncthreads();
  BnfLds L;
  L.frag = (const f32x4 *)bnf_lds;
  L.bias = (const f32x4 *)(bnf_lds + P.bias_off - P.lds_skip);
  L.norm = (const f32x4 *)(bnf_lds + P.norm_off - P.lds_skip);
  L.shift = (const int4 *)(bnf_lds + P.shift_off - P.lds_skip);
  float ev[3] = {0.0f, 0.0f, 0.0f};
  const int q = P.q;
  const int xcd = blockIdx.x & 7;
  // MODE 3: a unit is (item, state): both log posteriors of a Metropolis-Hastings iteration as independent evaluations in one launch
  const int n_units = MODE == 3 ? 2 * a.n_items : a.n_items;
  const int per = (n_units + 7) >> 3, lo = xcd * per, hi = min(n_units, lo + per);
  unsigned nacc_total = 0;
#ifdef BNF_PROF
  unsigned long long bnf_tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bnf_tl = __builtin_readcyclecounter();
  const unsigned long long bnf_t0 = bnf_tl;
#endif
  // Items are handed out by one counter per XCD (zeroed by the sign-word launch that precedes this one): a wave asks for its next
  // item when it starts the current one (the returning atomic is ~1 us, an item ~100 us).  With a static deal the four older waves
  // of a workgroup -- which win the SIMD's issue arbitration -- finished at 80 % of the kernel and left the younger four running
  // one per SIMD; the queue also trims the last round (15.26 items per wave on the bench panel).
  unsigned *queue = a.queue + xcd;
  unsigned tn = 0;
  if (lane == 0) tn = atomicAdd(queue, 1u);
  for (int unit = lo + (int)__builtin_amdgcn_readfirstlane(tn); unit < hi; unit = lo + (int)__builtin_amdgcn_readfirstlane(tn)) {
    if (lane == 0) tn = atomicAdd(queue, 1u);
    const int item = MODE == 3 ? unit >> 1 : unit, ust = MODE == 3 ? unit & 1 : 0;
    const int blk = item / a.groups_per_block, grp = item - blk * a.groups_per_block;
    const long long blk_lo = (long long)blk * a.bs;
    const int blk_n = (int)min((long long)a.bs, a.n - blk_lo);
    const int rib0 = grp * 16 * R;
    if (rib0 >= blk_n) continue;
    BNF_T(6);
    int rib[R];
    bool valid[R];
    float xr[R], yr[R];
    const float *xblk = a.x + blk_lo, *yblk = a.y + blk_lo;
    float *zblk = ((MODE == 3 && ust == 0) ? const_cast<float *>(a.zprop) : a.z) + blk_lo * q;
    const uint32_t rid0 = (uint32_t)(a.row_base + blk_lo);
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      const int r_ = rib0 + 16 * rt + j;
      valid[rt] = r_ < blk_n;
      rib[rt] = min(r_, blk_n - 1);
      xr[rt] = xblk[rib[rt]]; yr[rt] = yblk[rib[rt]];
    }
    const float *dwblk = a.dw + (long long)blk * a.n_states * P.set_floats;
    float zc[R][KS], zzc[R];
    if (MODE == 1 && a.init) {
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        const uint32_t rid = rid0 + (uint32_t)rib[rt];
#pragma unroll
        for (int sb = 0; sb < (KS + 3) / 4; ++sb) {
          const f32x4 e = box_muller4(philox4x32_10(rid, 0u, (uint32_t)(g + 4 * sb), TAG_INIT, a.k0, a.k1));
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * sb + r < KS) {
              const int f = 16 * sb + 4 * r + g;
              zc[rt][4 * sb + r] = f < q ? e[r] : 0.0f;
              if (f < q && valid[rt]) zblk[rib[rt] * q + f] = e[r];
            }
        }
      }
    } else {
#pragma unroll
      for (int rt = 0; rt < R; ++rt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int f = 16 * (ks >> 2) + 4 * (ks & 3) + g;
          const float t = zblk[rib[rt] * q + min(f, q - 1)];
          zc[rt][ks] = f < q ? t : 0.0f;
        }
    }
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      float s = 0.0f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s = fmaf(zc[rt][ks], zc[rt][ks], s);
      zzc[rt] = s;
    }
    if constexpr (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int rt = 0; rt < R; ++rt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          if (16 * (ks >> 2) + 4 * (ks & 3) + g == q) zc[rt][ks] = xr[rt];
      float lp[R], aux[R][3], lc0[R];
      if (a.prior) {
#pragma unroll
        for (int rt = 0; rt < R; ++rt)
          zzc[rt] = bnf_prior_share<KS>(a.prior + (long long)ust * a.prior_stride + (blk_lo + rib[rt]) * (long long)(q + 2), q, g, zc[rt], lc0[rt]);
      }
      const float *dws = dwblk + (long long)ust * P.set_floats;
      if constexpr (X3) bnx_logpost_rows<KS, R>(a, L, lane, j, g, blk_lo, rib, zc, xr, yr, dws, ust, zzc, lp BNF_PROF_ARG);
      else bnf_logpost_rows<KS, R, DET, WIDE>(a, L, lane, j, g, blk_lo, rib, zc, xr, yr, dws, ust, zzc, lp, aux BNF_PROF_ARG);
#pragma unroll
      for (int rt = 0; rt < R; ++rt)
        if (valid[rt] && g == 0) a.out[(long long)ust * a.n + blk_lo + rib[rt]] = a.prior ? lp[rt] - lc0[rt] : lp[rt];
      continue;
    }
    if constexpr (MODE == 2) {      // evaluate: reconstruction errors of g, h, f at the given latents
#pragma unroll
      for (int rt = 0; rt < R; ++rt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          if (16 * (ks >> 2) + 4 * (ks & 3) + g == q) zc[rt][ks] = xr[rt];
      float lp[R], aux[R][3];
      bnf_logpost_rows<KS, R, DET, WIDE, true>(a, L, lane, j, g, blk_lo, rib, zc, xr, yr, dwblk, 0, zzc, lp, aux BNF_PROF_ARG);
#pragma unroll
      for (int rt = 0; rt < R; ++rt)
        if (valid[rt] && g == 0) {
          const float xp = P.binary ? 1.0f / (1.0f + __expf(-aux[rt][1])) : aux[rt][1];
          const float dx = xr[rt] - xp, dy = yr[rt] - aux[rt][2];
          ev[0] += aux[rt][0]; ev[1] = fmaf(dx, dx, ev[1]); ev[2] = fmaf(dy, dy, ev[2]);
        }
      continue;
    }
    // proposal
    float zp[R][KS], zzp[R];
    const float sd = a.q_sd_blocks ? a.q_sd_blocks[blk] : a.q_sd;
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      const uint32_t rid = rid0 + (uint32_t)rib[rt];
      float s = 0.0f;
#pragma unroll
      for (int sb = 0; sb < (KS + 3) / 4; ++sb) {
        const f32x4 e = box_muller4(philox4x32_10(rid, (uint32_t)a.it, (uint32_t)(g + 4 * sb), TAG_PROP, a.k0, a.k1));
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * sb + r < KS) {
            const int f = 16 * sb + 4 * r + g;
            const float t = f < q ? fmaf(sd, e[r], zc[rt][4 * sb + r]) : 0.0f;
            zp[rt][4 * sb + r] = t;
            s = fmaf(t, t, s);
          }
      }
      zzp[rt] = s;
    }
    // the treatment rides in slot q of the extended input
#pragma unroll
    for (int rt = 0; rt < R; ++rt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        if (16 * (ks >> 2) + 4 * (ks & 3) + g == q) { zc[rt][ks] = xr[rt]; zp[rt][ks] = xr[rt]; }
    // both states through ONE copy of the network code (the kernel is ~25 KB of instructions instead of ~70): state 0 = proposal
    // (perturbation set 0, sign groups 0), state 1 = current.  Deterministic nets: the current state's value is cached per row
    // (evaluated here only when the chain is initialised).
    float lpp[R], lpc[R];
    float *lpblk = DET ? a.lp_cache + blk_lo : nullptr;
    const int st_hi = DET ? (a.init ? 2 : 1) : 2;
    if constexpr (DET) {
      if (!a.init) {
#pragma unroll
        for (int rt = 0; rt < R; ++rt) lpc[rt] = lpblk[rib[rt]];
      }
    }
#pragma nounroll
    for (int st = 0; st < st_hi; ++st) {
      float zs[R][KS], zzs[R], lp[R], aux[R][3], lcs[R];
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        zzs[rt] = st ? zzc[rt] : zzp[rt];
        lcs[rt] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) zs[rt][ks] = st ? zc[rt][ks] : zp[rt][ks];
      }
      if (a.prior) {      // the state's own call of the prior net (fresh noise per log-posterior evaluation, as for g, h, f)
#pragma unroll
        for (int rt = 0; rt < R; ++rt)
          zzs[rt] = bnf_prior_share<KS>(a.prior + (long long)st * a.prior_stride + (blk_lo + rib[rt]) * (long long)(q + 2), q, g, zs[rt], lcs[rt]);
      }
      if constexpr (X3) bnx_logpost_rows<KS, R>(a, L, lane, j, g, blk_lo, rib, zs, xr, yr, dwblk + (long long)st * P.set_floats, st, zzs, lp BNF_PROF_ARG);
      else bnf_logpost_rows<KS, R, DET, WIDE>(a, L, lane, j, g, blk_lo, rib, zs, xr, yr, DET ? nullptr : dwblk + (long long)st * P.set_floats, st, zzs,
                                              lp, aux BNF_PROF_ARG);
#pragma unroll
      for (int rt = 0; rt < R; ++rt) { if (st) lpc[rt] = lp[rt] - lcs[rt]; else lpp[rt] = lp[rt] - lcs[rt]; }
    }
    unsigned nacc = 0;
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      const uint4 w4 = philox4x32_10(rid0 + (uint32_t)rib[rt], (uint32_t)a.it >> 2, 0u, TAG_ACC, a.k0, a.k1);
      const int it = a.it;
      const unsigned w = (it & 2) ? ((it & 1) ? w4.w : w4.z) : ((it & 1) ? w4.y : w4.x);
      const bool acc = u01_open(w) < fast_exp(fminf(lpp[rt] - lpc[rt], 0.0f));
      if (valid[rt] && acc) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int f = 16 * (ks >> 2) + 4 * (ks & 3) + g;
          if (f < q) zblk[rib[rt] * q + f] = zp[rt][ks];
        }
        if (g == 0) ++nacc;
      }
      if constexpr (DET) {
        if (valid[rt] && g == 0 && (acc || a.init)) lpblk[rib[rt]] = acc ? lpp[rt] : lpc[rt];
      }
    }
    if (a.acc_count || a.acc_blocks) {
      for (int off = 32; off > 0; off >>= 1) nacc += __shfl_xor(nacc, off);
      if (lane == 0 && nacc) {
        if (a.acc_blocks) atomicAdd(&a.acc_blocks[blk], nacc);
        nacc_total += nacc;
      }
    }
  }
  if (a.acc_count && lane == 0 && nacc_total) atomicAdd(a.acc_count, nacc_total);
  if constexpr (MODE == 2) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float t = ev[k];
      for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
      if (lane == 0) atomicAdd(&a.sums[k], (double)t);
    }
  }
#ifdef BNF_PROF
  BNF_T(6);
  if (a.prof && tid == 0)
    for (int k = 0; k < 7; ++k) atomicAdd(&a.prof[k], bnf_tp[k]);
  if (a.prof && tid == 0) atomicAdd(&a.prof[28], bnf_tp[7]);
  if (a.prof && lane == 0) {      // every wave: its busy time (sum and maximum over the waves of the launch), by wave index
    const unsigned long long tot = __builtin_readcyclecounter() - bnf_t0;
    atomicAdd(&a.prof[7], tot);
    atomicMax(&a.prof[8], tot);
    atomicAdd(&a.prof[9 + wave], tot);
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// causal effects: the outcome net at n_doses treatments on one state per row (infer_from_latent_posterior :671-763, one kept
// draw per launch).  Dose k is its own noisy call (perturbation set k of the block, sign groups k of the row).
// ---------------------------------------------------------------------------------------------
struct BnfEffArgs {
  BnfPlan pl;
  const float *eblob;                  // [e_blob_floats]: outcome net only
  const float *dw;                     // [n_blocks][n_doses][e_frags * 256]
  const uint4 *sgf;                    // [n_doses][BNF_NG_H][n]
  const float *z;                      // [n x q]
  long long n, row_base;
  int bs, n_blocks, block0, groups_per_block, n_items, n_doses;
  const float *xvals;
  uint32_t k0, k1;
  int sample_y;
  uint32_t it_noise;
  double *sum_out; long long sum_stride;
  float *ite_out; long long ite_stride;
  unsigned *queue;                     // [8] item counters, one per XCD (zeroed by bnf_signs_kernel)
  float *fsum_out; long long fsum_slot_stride;   // deterministic API: per-workgroup partial sums [grid][...] += (bgm_adrf_reduce adds the slots)
  float sig2_y;                        // fixed params['sigma_y']^2 (<= 0: the variance head)
};

template <int KS, int R, int WAVES, bool DET = false, bool X3 = false>      // KS: k-steps of the outcome net's own first layer (BnfPlan::KSF or a larger compiled value)
static __global__ __launch_bounds__(64 * WAVES, WAVES / 4) void bnf_effects_kernel(BnfEffArgs a) {
  static_assert(!X3 || !DET, "bnf_effects_kernel: split precision serves the Bayesian nets");
  extern __shared__ __attribute__((aligned(16))) float bnf_lds[];
  constexpr int T0 = (KS + 3) / 4;
  const BnfPlan &P = a.pl;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float *acc_lds = bnf_lds + ((P.e_blob_floats + 3) & ~3);     // [WAVES][BNF_MAX_DOSES]
  {
    const f32x4 *src = (const f32x4 *)a.eblob;
    f32x4 *dst = (f32x4 *)bnf_lds;
    const int cnt = P.e_blob_floats >> 2;
    for (int i = tid; i < cnt; i += 64 * WAVES) dst[i] = src[i];
    for (int i = tid; i < WAVES * BNF_MAX_DOSES; i += 64 * WAVES) acc_lds[i] = 0.0f;
  }
  __syncthreads();
  const f32x4 *LF = (const f32x4 *)bnf_lds;
  const f32x4 *BL = (const f32x4 *)(bnf_lds + P.e_bias_off);
  const f32x4 *NORM = (const f32x4 *)(bnf_lds + P.e_norm_off);
  const int4 *SHIFT = (const int4 *)(bnf_lds + P.e_shift_off);
  float *myacc = acc_lds + wave * BNF_MAX_DOSES;
  const int q = P.q, nd = a.n_doses, zz = P.z0 + P.z1;
  const long long eset = (long long)P.e_frags * 256;
  const int xcd = blockIdx.x & 7;
  const int per = (a.n_items + 7) >> 3, lo = xcd * per, hi = min(a.n_items, lo + per);
  unsigned *queue = a.queue + xcd;
  unsigned tn = 0;
  if (lane == 0) tn = atomicAdd(queue, 1u);
  for (int item = lo + (int)__builtin_amdgcn_readfirstlane(tn); item < hi; item = lo + (int)__builtin_amdgcn_readfirstlane(tn)) {
    if (lane == 0) tn = atomicAdd(queue, 1u);
    const int blk = item / a.groups_per_block, grp = item - blk * a.groups_per_block;
    const long long blk_lo = (long long)blk * a.bs;
    const int blk_n = (int)min((long long)a.bs, a.n - blk_lo);
    const int rib0 = grp * 16 * R;
    if (rib0 >= blk_n) continue;
    int rib[R];
    bool valid[R];
    float ze[R][KS];
    const float *zblk = a.z + blk_lo * q;
    const uint4 *SGb = a.sgf + blk_lo;
    const uint32_t rid0 = (uint32_t)(a.row_base + blk_lo);
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      const int r_ = rib0 + 16 * rt + j;
      valid[rt] = r_ < blk_n;
      rib[rt] = min(r_, blk_n - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int f = 16 * (ks >> 2) + 4 * (ks & 3) + g;
        const float t = zblk[rib[rt] * q + min(f, max(zz, 1) - 1)];
        ze[rt][ks] = f < zz ? t : 0.0f;
      }
    }
    const float *dwblk = a.dw + (long long)blk * nd * eset;
    float y0[R];
    f32x4 nz[R];
#pragma unroll
    for (int rt = 0; rt < R; ++rt) { y0[rt] = 0.0f; nz[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    uint4 Gn[R][BNF_NG_H];
    if constexpr (!DET) {
#pragma unroll
      for (int rt = 0; rt < R; ++rt)
#pragma unroll
        for (int k = 0; k < BNF_NG_H; ++k) Gn[rt][k] = (SGb + (long long)k * a.n)[rib[rt]];
    }
#pragma nounroll
    for (int k = 0; k < nd; ++k) {
      const float xv = a.xvals[k];
      uint4 G[R][BNF_NG_H];
      if constexpr (!DET) {
#pragma unroll
        for (int rt = 0; rt < R; ++rt)
#pragma unroll
          for (int c = 0; c < BNF_NG_H; ++c) G[rt][c] = Gn[rt][c];
      }
      if (!DET && k + 1 < nd) {
        const uint4 *S = SGb + (long long)(k + 1) * BNF_NG_H * a.n;
#pragma unroll
        for (int rt = 0; rt < R; ++rt)
#pragma unroll
          for (int c = 0; c < BNF_NG_H; ++c) Gn[rt][c] = (S + (long long)c * a.n)[rib[rt]];
      }
      BNF_PIN();
      // outcome noise: lane group gg draws call 4 (k >> 4) + gg once per 16 doses; dose k is finished by lane group (k >> 2) & 3
      if (a.sample_y && (k & 15) == 0) {
#pragma unroll
        for (int rt = 0; rt < R; ++rt)
          nz[rt] = box_muller4(philox4x32_10(rid0 + (uint32_t)rib[rt], a.it_noise, (uint32_t)((k >> 2) + g), TAG_YNOISE, a.k0, a.k1));
      }
#pragma unroll
      for (int rt = 0; rt < R; ++rt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          if (16 * (ks >> 2) + 4 * (ks & 3) + g == zz) ze[rt][ks] = xv;
      float mu[R], raw[R];
      BGM_NO_HOIST();
      if constexpr (X3) bnx_head<KS, R>(LF, (const f32x4 *)(dwblk + (long long)k * eset), BL, NORM, SHIFT, lane, g, ze, G, mu, raw);
      else bnf_head<KS, R, DET>(LF, (const f32x4 *)(dwblk + (long long)k * eset), BL, NORM, SHIFT, lane, g, ze, G, mu, raw);
      const int own = (k >> 2) & 3, e = k & 3;
      float tot = 0.0f;
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        float yk = mu[rt];
        if (a.sample_y) {
          const float s2 = (a.sig2_y > 0.0f) ? a.sig2_y : bnf_softplus(raw[rt]) + BGM_EPS;
          yk = fmaf(__builtin_sqrtf(s2), e == 0 ? nz[rt][0] : e == 1 ? nz[rt][1] : e == 2 ? nz[rt][2] : nz[rt][3], yk);
        }
        if (a.ite_out) {
          if (k == 0) y0[rt] = yk;
          else if (k == 1 && valid[rt] && g == own) a.ite_out[(blk_lo + rib[rt]) * a.ite_stride] = y0[rt] - yk;
        }
        tot += valid[rt] ? yk : 0.0f;
      }
      if (a.sum_out || a.fsum_out) {
        tot = sum_over_j_to_lane15(tot);
        if (lane == 16 * own + 15) myacc[k] += tot;
      }
    }
  }
  if (a.sum_out || a.fsum_out) {
    __syncthreads();
    for (int k = tid; k < nd; k += 64 * WAVES) {
      double t = 0.0;
      for (int w = 0; w < WAVES; ++w) t += (double)acc_lds[w * BNF_MAX_DOSES + k];
      if (a.sum_out) atomicAdd(&a.sum_out[(long long)k * a.sum_stride], t);
      else a.fsum_out[(long long)blockIdx.x * a.fsum_slot_stride + (long long)k * a.sum_stride] += (float)t;      // this workgroup's slot
    }
  }
}

// ---------------------------------------------------------------------------------------------
// packing, perturbations, sign words
// ---------------------------------------------------------------------------------------------
// Element tables (host-built, bnf_api.hip).  A kernel element: theta offsets of loc / rho, position in the set, replication.
struct BnfWElem { int loc, rho, pos, rep; float scale; int posx; };     // rep copies at pos, pos + 16, ...; scale 0.6 behind a LeakyReLU; posx: the split layout's position word (bnx_kernels.h)
struct BnfBElem { int src, pos, rep; };                       // bias -> bias tiles (float offset from bias_off)
struct BnfNElem { int gamma, beta, pos_sc, pos_sh, pos_shift, shift; };   // one slot of one net's extended input (gamma < 0: unused)

struct BnfPackArgs {
  const float *theta;
  const BnfWElem *w; int n_w;
  const BnfBElem *b; int n_b;
  const BnfNElem *ne; int n_n;
  float *blob, *sf;           // blob [frags | bias tiles | norm | shift]; sigma fragments [n_frags * 256]
  int bias_off;
};
static __global__ void bnf_pack_kernel(BnfPackArgs a) {
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x, str = gridDim.x * blockDim.x;
  for (int i = i0; i < a.n_w; i += str) {
    const BnfWElem e = a.w[i];
    const float lv = e.scale * a.theta[e.loc], sv = e.rho >= 0 ? e.scale * (BNN_SCALE_EPS + softplus_acc(a.theta[e.rho])) : 0.0f;
    for (int c = 0; c < e.rep; ++c) { a.blob[e.pos + 16 * c] = lv; a.sf[e.pos + 16 * c] = sv; }
  }
  for (int i = i0; i < a.n_b; i += str) {
    const BnfBElem e = a.b[i];
    const float v = a.theta[e.src];
    for (int c = 0; c < e.rep; ++c) a.blob[a.bias_off + e.pos + 4 * c] = v;
  }
  for (int i = i0; i < a.n_n; i += str) {
    const BnfNElem e = a.ne[i];
    // gamma >= 0: inference-mode BatchNormalization of a used slot; -2: a used slot of a net without input normalisation; -1: unused
    a.blob[e.pos_sc] = e.gamma >= 0 ? a.theta[e.gamma] / sqrtf(1.0f + BNN_BN_EPS) : (e.gamma == -2 ? 1.0f : 0.0f);
    a.blob[e.pos_sh] = e.gamma >= 0 ? a.theta[e.beta] : 0.0f;
    ((int *)a.blob)[e.pos_shift] = e.shift;
  }
}

// dW = sigma * eps of every (block, state): grid (chunks, n_blocks * n_states).  Layer descriptors give the canonical element
// range of each Flipout kernel; element idx of a layer belongs to Philox call idx >> 2 (oracle/bnn.py draw_noise).  All layers of
// a set form ONE index space of Philox calls (c_base = calls before the layer) so that every thread of the launch has work; the
// positions come from a 4-byte table (position | (copies - 1) << 28).
struct BnfLayerDesc { int e_base, cnt, l, net_id, c_base; };
struct BnfNoiseArgs {
  BnfLayerDesc lay[14];
  int n_lay, n_calls;
  const int *npos;            // [elements] position in the set, copies - 1 in the top nibble
  const float *sf;            // sigma fragments in the layout of the position table
  float *dw; long long set_floats;
  int n_states;
  uint32_t k0, k1, stream0;
  int block0;
};
static __global__ __launch_bounds__(256) void bnf_noise_kernel(BnfNoiseArgs a) {
  const int set = blockIdx.y, blk = set / a.n_states, s = set - blk * a.n_states;
  const uint32_t k1 = a.k1 + (uint32_t)(a.block0 + blk), stream = a.stream0 + (uint32_t)s;
  float *dw = a.dw + (long long)set * a.set_floats;
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
      if (idx < L.cnt) {
        const int pe = a.npos[L.e_base + idx], pos = pe & 0x0FFFFFFF, rep = (pe >> 28) & 15;
        const float v = a.sf[pos] * z[u];
        for (int r = 0; r <= rep; ++r) dw[pos + 16 * r] = v;
      }
    }
  }
}

// Rademacher sign words of every (row, state) in the layer-aligned groups the sampler loads.  nets: bit 0 g, 1 h, 2 f.
struct BnfSignArgs {
  uint4 *g; uint32_t *gout; uint4 *h, *f;
  long long n;
  int bs, block0, n_states, nets;
  uint32_t k0, k1, stream0;
  unsigned *queue;            // [8] item counters of the sampler / effects launch that follows: cleared here
  long long rib0;             // position of this call's first row inside its block (a rank's share of ONE block: IdentifiableCausalBGM.predict)
};
static __global__ __launch_bounds__(256) void bnf_signs_kernel(BnfSignArgs a) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  const int s = blockIdx.y;
  if (blockIdx.x == 0 && s == 0 && threadIdx.x < 8) a.queue[threadIdx.x] = 0u;
  if (row >= a.n) return;
  const int blk = (int)((a.rib0 + row) / a.bs);
  const uint32_t rib = (uint32_t)(a.rib0 + row - (long long)blk * a.bs);      // the words are keyed by (block, position in the block)
  const uint32_t k1 = a.k1 + (uint32_t)(a.block0 + blk), stream = a.stream0 + (uint32_t)s;
  if (a.nets & 1) {
    uint32_t w[28];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
      const uint4 t = philox4x32_10(rib, (uint32_t)c | ((uint32_t)BNN_G << 16), stream, BNN_TAG_SIGN, a.k0, k1);
      w[4 * c] = t.x; w[4 * c + 1] = t.y; w[4 * c + 2] = t.z; w[4 * c + 3] = t.w;
    }
    uint4 *G = a.g + (long long)s * BNF_NG_G * a.n + row;
    G[0] = make_uint4(w[0], w[1], w[2], 0u);
#pragma unroll
    for (int l = 1; l <= 4; ++l) G[(long long)l * a.n] = make_uint4(w[4 * l - 1], w[4 * l], w[4 * l + 1], w[4 * l + 2]);
    G[5LL * a.n] = make_uint4(w[19], w[20], 0u, 0u);
    uint4 *O = (uint4 *)(a.gout + ((long long)s * a.n + row) * BNF_GOUT);
    O[0] = make_uint4(w[21], w[22], w[23], w[24]);
    O[1] = make_uint4(w[25], w[26], w[27], 0u);
  }
#pragma unroll
  for (int hn = 0; hn < 2; ++hn) {
    if (!(a.nets & (2 << hn))) continue;
    const uint32_t net_id = hn == 0 ? (uint32_t)BNN_H : (uint32_t)BNN_F;
    uint32_t w[12];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint4 t = philox4x32_10(rib, (uint32_t)c | (net_id << 16), stream, BNN_TAG_SIGN, a.k0, k1);
      w[4 * c] = t.x; w[4 * c + 1] = t.y; w[4 * c + 2] = t.z; w[4 * c + 3] = t.w;
    }
    uint4 *H = (hn == 0 ? a.h : a.f) + (long long)s * BNF_NG_H * a.n + row;
    H[0] = make_uint4(w[0], w[1], w[2], 0u);
    H[a.n] = make_uint4(w[3], w[4], w[5], w[6]);
    H[2LL * a.n] = make_uint4(w[7], w[8], w[9], 0u);
  }
}
