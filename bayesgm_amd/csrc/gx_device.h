// gx_device.h -- the general-width MLP engine of the gfx950 library ("gx"): dense layers of ANY width and depth on the fp32 matrix
// pipe, for the models whose hidden layers are not the reference defaults (g / e [64] x k, f / h [64, 32, 8]) that the resident and
// streamed-fragment kernel families are compiled for.
//
// replaces: BaseFullyConnectedNet / BaseVariationalNet of models/networks/base.py:4-117 with arbitrary `nb_units`
// (networks/base.py:7 default [256, 256, 256]; r-package/bayesgm/tests/testthat/test-causalbgm.R:28-34 uses (8, 8) / (8, 4)).
//
// Execution layout.  A workgroup of 4 waves owns a tile of 32 rows (observations / chains).  Activations live in LDS, row-major
// [32][ld] with ld = 8 (mod 64) floats -- the stride at which the 16-byte A-operand reads below are bank-conflict free on gfx950
// (ds_read_b128 is serviced in four 16-lane groups, MI355X_MICROARCH.md "LDS").  Weights stay in HBM / L2 in a PADDED copy of the
// Keras-order parameters: every layer W [K][N] row-major with K and N rounded up to multiples of 32 and zero filled, so no inner
// loop carries a bound check and padded features are exact zeros end to end (LeakyReLU(0) = 0, zero rows / columns / biases).
// The backward products read a second, transposed padded copy (W^T [N][K]) through the same routine.
//
// One layer  Y[32 x N] = A[32 x K] W[K x N]  is dealt to the waves in units of (16-row tile, 32-column group).  A unit contracts K in
// blocks of 16 on v_mfma_f32_16x16x4_f32 (exact fp32 multiply-add chains, M = rows, N = output features): lane (j = lane & 15,
// g = lane >> 4) reads ONE 16-byte A fragment A[row j][k0 + 4g .. + 3] from LDS and four 8-byte weight pairs
// W[k0 + 4g + s][n0 + 2j, n0 + 2j + 1] (coalesced 128-byte row segments) per block and issues 8 MFMAs; K-step s contracts over
// {k0 + 4g + s : g} on both operands.  The two accumulators of a lane hold columns n0 + 2j and n0 + 2j + 1 of rows 16 rt + 4g + r, so
// an epilogue stores 8-byte pairs.  Per 8 MFMAs (256 matrix-pipe cycles) a wave issues 5 loads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bgm_device.h"

#define GX_THREADS 256
#define GX_WAVES 4
#define GX_ROWS 32
#define GX_MAXL 9           // dense layers of one network: up to BGM_MAX_LAYERS hidden + the output layer

struct GxNet {              // one fully connected network inside the padded packs
  int L;                    // dense layers
  int dim[GX_MAXL + 1];     // true widths  [in, h1, ..., out]
  int pad[GX_MAXL + 1];     // padded widths (multiples of 32)
  int w[GX_MAXL];           // W_l  [pad[l]][pad[l+1]]   at pack + w[l]
  int b[GX_MAXL];           // b_l  [pad[l+1]]           at pack + b[l]
  int wt[GX_MAXL];          // W_l^T [pad[l+1]][pad[l]]  at packT + wt[l]
  int wx[GX_MAXL];          // W_l as hi / lo fp16 fragments at packx + wx[l] (dwords; gx_dense_x3)
  int base;                 // first canonical (Keras-order) parameter of the net in the model's flat parameter vector
};

__host__ __device__ inline int gx_pad32(int n) { return (n + 31) & ~31; }
// LDS row stride for activations of up to `width` (padded) floats: the smallest ld >= width + 4 with ld = 8 (mod 64)
__host__ __device__ inline int gx_ld(int width) { return ((width + 59) / 64) * 64 + 8; }

// ---------------------------------------------------------------------------------------------------------------------------
// Y = A W for the workgroup's 32 rows.  W [K][N] (padded, global), A [32][lda] in LDS (A_GLOBAL = false) or in global memory
// (true: row r of the tile at A + r * lda); gx_dense_ld: the N columns processed are a slice of rows of stride ldw.  epi(rt, n0, acc0, acc1): lane (j, g) holds rows 16 rt + 4g + r (r = 0..3) of columns
// n0 + 2j (acc0[r]) and n0 + 2j + 1 (acc1[r]).  No barriers inside: the caller separates producers and consumers of A.
// ---------------------------------------------------------------------------------------------------------------------------
// First K block of the wave's FIRST unit of a layer (four 8-byte weight pairs) and the unit's bias pair, requested ahead of the barrier
// that releases the layer's input (weights and biases do not depend on it): the L2 round trip of every layer's first block otherwise
// sits in front of its first MFMA (measured: ~1.5-3 k cycles per layer of a 64-wide net, more than the layer's own 1 k of matrix work).
struct GxPre { f32x2 b0, b1, b2, b3, bb; int valid; };
// the lane index behind a compiler-opaque copy: everything derived from it (fragment offsets of every layer) is then NOT invariant of the
// surrounding iteration loop -- hoisted out of it, those offsets of all layers stay live across a whole transition and spill
__device__ __forceinline__ int gx_lane() { int l = threadIdx.x & 63; asm volatile("" : "+v"(l)); return l; }
// (u0 >= 0: the calling wave's first unit instead of unit `wave` -- the row-tile-per-wave kernels of gw_kernels.h walk all units themselves)
// FRAG: W is a layer of the FRAGMENT pack (gx_api.hip): per (32-column group cg, 16-row K block kb) the 64 lanes' operands of the block
// are contiguous -- lane (j, g) owns 8 floats {W[16 kb + 4 g + s][32 cg + 2 j + c]: s = 0..3, c = 0, 1} at ((cg * K / 16 + kb) * 64 + lane) * 8:
// two 16-byte requests per lane and block (one contiguous 2 KB per wave) instead of four 8-byte ones on four rows, one address step.
template <bool FRAG = false>
__device__ __forceinline__ GxPre gx_prefetch(const float *__restrict__ W, int ldw, int N, const float *bias, int nrt = 2, int u0 = -1, int K = 0) {
  GxPre p;
  p.valid = 0;
  const int lane = gx_lane(), wave = u0 >= 0 ? u0 : (int)(threadIdx.x >> 6), j = lane & 15, g = lane >> 4;
  if (wave >= nrt * (N >> 5)) return p;
  const int n0 = (wave / nrt) << 5;
  if constexpr (FRAG) {
    const f32x4 *wf = reinterpret_cast<const f32x4 *>(W + ((size_t)(n0 >> 5) * (K >> 4) * 64 + lane) * 8);
    const f32x4 q0 = wf[0], q1 = wf[1];
    p.b0 = f32x2{q0[0], q0[1]}; p.b1 = f32x2{q0[2], q0[3]}; p.b2 = f32x2{q1[0], q1[1]}; p.b3 = f32x2{q1[2], q1[3]};
  } else {
  const float *wk = W + (size_t)(4 * g) * ldw + n0 + 2 * j;
  p.b0 = *reinterpret_cast<const f32x2 *>(wk);
  p.b1 = *reinterpret_cast<const f32x2 *>(wk + ldw);
  p.b2 = *reinterpret_cast<const f32x2 *>(wk + 2 * (size_t)ldw);
  p.b3 = *reinterpret_cast<const f32x2 *>(wk + 3 * (size_t)ldw);
  }
  p.bb = bias ? *reinterpret_cast<const f32x2 *>(bias + n0 + 2 * j) : f32x2{0.0f, 0.0f};
  p.valid = 1;
  return p;
}
// optional hooks of an epilogue functor: pre(rt, n0) = requests of unit (rt, n0)'s epilogue operands, issued one unit ahead (with the
// unit's first weight block); rotate() = behind each epilogue call (next unit's operands become the current ones)
template <class E> __device__ __forceinline__ auto gx_epi_pre(E &e, int rt, int n0, int) -> decltype(e.pre(rt, n0), void()) { e.pre(rt, n0); }
template <class E> __device__ __forceinline__ void gx_epi_pre(E &, int, int, long) {}
template <class E> __device__ __forceinline__ auto gx_epi_rotate(E &e, int) -> decltype(e.rotate(), void()) { e.rotate(); }
template <class E> __device__ __forceinline__ void gx_epi_rotate(E &, long) {}

// nrt: 16-row tiles of A (2 = the workgroup's 32 rows; the effect pass stacks several doses' rows: 2 x doses).  bias != NULL: the
// accumulators start from the unit's bias pair (requested with its first weight block; the functor then adds none).  pre: the first
// unit's first block as requested by gx_prefetch ahead of the barrier in front of this call.
// u0 / ustride: the calling wave's first unit and its stride over the units (default: unit `wave`, stride GX_WAVES -- the workgroup's
// waves deal the units; 0 / 1: one wave walks them all on ITS rows A, gw_kernels.h).
template <bool A_GLOBAL = false, bool FRAG = false, class Epi>
__device__ __forceinline__ void gx_dense_ld(const float *__restrict__ W, int ldw, int K, int N, const float *A, int lda, Epi epi, int nrt = 2,
                                            const float *bias = nullptr, const GxPre *pre = nullptr, int u0 = -1, int ustride = GX_WAVES) {
  const int lane = gx_lane(), wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int units = nrt * (N >> 5);
  int u = u0 >= 0 ? u0 : wave;
  if (u >= units) return;
  // weights are addressed as (uniform base W) + (32-bit byte offset per lane): four row offsets, each advanced by one addition per K
  // block (64-bit per-lane pointers cost 8 v_lshl_add_u64 + 6 moves per block; the packs stay below 2^30 floats, gx_api.hip)
  const unsigned wstep = 64u * (unsigned)ldw;                                         // 16 rows of W in bytes
  auto wld = [&](unsigned boff) { return *reinterpret_cast<const f32x2 *>(reinterpret_cast<const char *>(W) + boff); };
  // software pipeline over K blocks AND over the wave's units: the operands of K block k0 + 16 are requested before the MFMAs of
  // block k0 issue, and block 0 of the wave's NEXT unit before the MFMAs of the current unit's last block -- a unit of a 64-wide
  // layer is 4 blocks (1024 matrix cycles), and starting each one on a cold L2 round trip cost more than the unit itself
  int rt = u % nrt, n0 = (u / nrt) << 5;
  const float *ap = A + (size_t)(16 * rt + j) * lda + 4 * g;
  const unsigned ldb = 4u * (unsigned)ldw;
  // FRAG: one byte offset, 2 KB per K block, two 16-byte requests (see gx_prefetch)
  const unsigned ubytes = 2048u * (unsigned)(K >> 4);                                 // one column group's blocks in the fragment pack
  auto fld = [&](unsigned boff, f32x2 &r0, f32x2 &r1, f32x2 &r2, f32x2 &r3) {
    const f32x4 *wf = reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(W) + boff);
    const f32x4 q0 = wf[0], q1 = wf[1];
    r0 = f32x2{q0[0], q0[1]}; r1 = f32x2{q0[2], q0[3]}; r2 = f32x2{q1[0], q1[1]}; r3 = f32x2{q1[2], q1[3]};
  };
  unsigned o0 = FRAG ? (unsigned)(n0 >> 5) * ubytes + 32u * (unsigned)lane : 4u * ((unsigned)(4 * g) * (unsigned)ldw + (unsigned)(n0 + 2 * j));
  unsigned o1 = o0 + ldb, o2 = o1 + ldb, o3 = o2 + ldb;
  f32x4 a = *reinterpret_cast<const f32x4 *>(ap);
  f32x2 b0, b1, b2, b3, bb = {0.0f, 0.0f};
  if (pre != nullptr && pre->valid) { b0 = pre->b0; b1 = pre->b1; b2 = pre->b2; b3 = pre->b3; bb = pre->bb; }
  else {
    if constexpr (FRAG) fld(o0, b0, b1, b2, b3);
    else { b0 = wld(o0); b1 = wld(o1); b2 = wld(o2); b3 = wld(o3); }
    if (bias) bb = *reinterpret_cast<const f32x2 *>(bias + n0 + 2 * j);
  }
  gx_epi_pre(epi, rt, n0, 0);
  gx_epi_rotate(epi, 0);
  for (;;) {
    f32x4 acc0 = {bb[0], bb[0], bb[0], bb[0]}, acc1 = {bb[1], bb[1], bb[1], bb[1]};
    // (two K blocks per trip with the two operand sets exchanging roles by NAME: the rotation a = an, b = c of a one-block body is ten
    // register moves per block that also need the requested data inside the issuing iteration)
    int k0 = 16;
    for (; k0 + 16 < K; k0 += 32) {
      f32x2 c0, c1, c2, c3;
      const f32x4 an = *reinterpret_cast<const f32x4 *>(ap + k0);
      if constexpr (FRAG) { o0 += 2048u; fld(o0, c0, c1, c2, c3); }
      else { o0 += wstep; o1 += wstep; o2 += wstep; o3 += wstep; c0 = wld(o0); c1 = wld(o1); c2 = wld(o2); c3 = wld(o3); }
      __builtin_amdgcn_sched_barrier(0);      // keep the requests above the MFMAs of the previous block (hipcc sinks loads to their first use)
      acc0 = BGM_MFMA(a[0], b0[0], acc0); acc1 = BGM_MFMA(a[0], b0[1], acc1);
      acc0 = BGM_MFMA(a[1], b1[0], acc0); acc1 = BGM_MFMA(a[1], b1[1], acc1);
      acc0 = BGM_MFMA(a[2], b2[0], acc0); acc1 = BGM_MFMA(a[2], b2[1], acc1);
      acc0 = BGM_MFMA(a[3], b3[0], acc0); acc1 = BGM_MFMA(a[3], b3[1], acc1);
      a = *reinterpret_cast<const f32x4 *>(ap + k0 + 16);
      if constexpr (FRAG) { o0 += 2048u; fld(o0, b0, b1, b2, b3); }
      else { o0 += wstep; o1 += wstep; o2 += wstep; o3 += wstep; b0 = wld(o0); b1 = wld(o1); b2 = wld(o2); b3 = wld(o3); }
      __builtin_amdgcn_sched_barrier(0);
      acc0 = BGM_MFMA(an[0], c0[0], acc0); acc1 = BGM_MFMA(an[0], c0[1], acc1);
      acc0 = BGM_MFMA(an[1], c1[0], acc0); acc1 = BGM_MFMA(an[1], c1[1], acc1);
      acc0 = BGM_MFMA(an[2], c2[0], acc0); acc1 = BGM_MFMA(an[2], c2[1], acc1);
      acc0 = BGM_MFMA(an[3], c3[0], acc0); acc1 = BGM_MFMA(an[3], c3[1], acc1);
    }
    if (k0 < K) {                              // an odd block left
      f32x2 c0, c1, c2, c3;
      const f32x4 an = *reinterpret_cast<const f32x4 *>(ap + k0);
      if constexpr (FRAG) { o0 += 2048u; fld(o0, c0, c1, c2, c3); }
      else { o0 += wstep; o1 += wstep; o2 += wstep; o3 += wstep; c0 = wld(o0); c1 = wld(o1); c2 = wld(o2); c3 = wld(o3); }
      __builtin_amdgcn_sched_barrier(0);
      acc0 = BGM_MFMA(a[0], b0[0], acc0); acc1 = BGM_MFMA(a[0], b0[1], acc1);
      acc0 = BGM_MFMA(a[1], b1[0], acc0); acc1 = BGM_MFMA(a[1], b1[1], acc1);
      acc0 = BGM_MFMA(a[2], b2[0], acc0); acc1 = BGM_MFMA(a[2], b2[1], acc1);
      acc0 = BGM_MFMA(a[3], b3[0], acc0); acc1 = BGM_MFMA(a[3], b3[1], acc1);
      a = an; b0 = c0; b1 = c1; b2 = c2; b3 = c3;
    }
    // last block of this unit; block 0 of the next one is requested first
    const int un = u + ustride;
    const bool more = un < units;
    const int rtn = un % nrt, n0n = (un / nrt) << 5;
    f32x4 an = a;
    f32x2 c0 = b0, c1 = b1, c2 = b2, c3 = b3, bn = bb;
    if (more) {
      ap = A + (size_t)(16 * rtn + j) * lda + 4 * g;
      an = *reinterpret_cast<const f32x4 *>(ap);
      if constexpr (FRAG) { o0 = (unsigned)(n0n >> 5) * ubytes + 32u * (unsigned)lane; fld(o0, c0, c1, c2, c3); }
      else {
        o0 = 4u * ((unsigned)(4 * g) * (unsigned)ldw + (unsigned)(n0n + 2 * j)); o1 = o0 + ldb; o2 = o1 + ldb; o3 = o2 + ldb;
        c0 = wld(o0); c1 = wld(o1); c2 = wld(o2); c3 = wld(o3);
      }
      if (bias) bn = *reinterpret_cast<const f32x2 *>(bias + n0n + 2 * j);
      gx_epi_pre(epi, rtn, n0n, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    acc0 = BGM_MFMA(a[0], b0[0], acc0); acc1 = BGM_MFMA(a[0], b0[1], acc1);
    acc0 = BGM_MFMA(a[1], b1[0], acc0); acc1 = BGM_MFMA(a[1], b1[1], acc1);
    acc0 = BGM_MFMA(a[2], b2[0], acc0); acc1 = BGM_MFMA(a[2], b2[1], acc1);
    acc0 = BGM_MFMA(a[3], b3[0], acc0); acc1 = BGM_MFMA(a[3], b3[1], acc1);
    epi(rt, n0, acc0, acc1);
    if (!more) break;
    gx_epi_rotate(epi, 0);
    u = un; rt = rtn; n0 = n0n;
    a = an; b0 = c0; b1 = c1; b2 = c2; b3 = c3; bb = bn;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Split precision ("f16 x 3", opt-in: bgm_causal_set_precision(2) on the row-tile-per-wave kernels of gw_kernels.h).  Y = A W for the
// wave's nrt row tiles with every fp32 operand as hi + lo fp16 and three products W_hi A_hi + W_hi A_lo + W_lo A_hi on
// v_mfma_f32_16x16x32_f16 (16 cycles for K = 32 against 8 x 32 cycles of the fp32 instruction), fp32 accumulation from the bias.
// The accumulators come out in gx_dense_ld's layout (lane (j, g): rows 4 g + r of columns n0 + 2 j | n0 + 2 j + 1), so every epilogue
// functor serves both.  Activations stay fp32 in LDS: a lane reads its row's 8 k-values of a K block (k = 32 kb + 8 g + i) and splits
// them in registers, ONCE per row tile for all column groups (K <= 128 on these kernels).
// WX: the layer of the split pack (gx_api.hip): per (32-column group cg, K block kb of 32) and lane 16 dwords =
// [hi of column 2 j | hi of column 2 j + 1 | lo of 2 j | lo of 2 j + 1], each 8 fp16 (k = 32 kb + 8 g + 0 .. 7): four 16-byte requests.
typedef _Float16 gx_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 gx_h2 __attribute__((ext_vector_type(2)));
typedef float gx_f2 __attribute__((ext_vector_type(2)));
typedef unsigned gx_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gx_split_pair(float a, float b, unsigned &hi, unsigned &lo) {
  const gx_h2 h = __builtin_convertvector(gx_f2{a, b}, gx_h2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(gx_f2{a - (float)h[0], b - (float)h[1]}, gx_h2));
}
__device__ __forceinline__ void gx_split8(const f32x4 &a, const f32x4 &b, gx_u4 &hi, gx_u4 &lo) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  gx_split_pair(a[0], a[1], h0, l0); gx_split_pair(a[2], a[3], h1, l1);
  gx_split_pair(b[0], b[1], h2, l2); gx_split_pair(b[2], b[3], h3, l3);
  hi = gx_u4{h0, h1, h2, h3}; lo = gx_u4{l0, l1, l2, l3};
}
#define GX_MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(gx_h8, a), __builtin_bit_cast(gx_h8, b), c, 0, 0, 0)
#define GX_X3_MAXKB 4      // K <= 128
// first K block of a layer's first unit, requested ahead of the layer in front of it (as GxPre for the fp32 routine)
struct GxPreX { gx_u4 w0, w1, w2, w3; int valid; };
__device__ __forceinline__ GxPreX gx_prefetch_x3(const unsigned *__restrict__ WX) {
  GxPreX p;
  const gx_u4 *wp = reinterpret_cast<const gx_u4 *>(WX) + (size_t)gx_lane() * 4;
  p.w0 = wp[0]; p.w1 = wp[1]; p.w2 = wp[2]; p.w3 = wp[3]; p.valid = 1;
  return p;
}
// The wave's walk is ONE stream of K blocks over the units (block b = unit * KB + kb at WX + b * 4 KB for nrt = 1; with several row
// tiles a column group's blocks are walked once per tile): block b + 1 is requested before the products of block b, across unit
// boundaries -- a unit is 6 KB products (a few hundred cycles), an L2 round trip in front of each would cost more than the unit.
// MAXKB: K blocks whose split operands live in registers at a time (8 registers each).  RESPLIT = false: K <= 32 MAXKB, the input is
// split once per row tile for all column groups; true: any K up to 128 in chunks of MAXKB blocks, re-split per unit (more conversions,
// 16 registers instead of 32 at MAXKB = 2: a fourth wave per SIMD).
template <int MAXKB = GX_X3_MAXKB, bool RESPLIT = false, class Epi>
__device__ __forceinline__ void gx_dense_x3(const unsigned *__restrict__ WX, int K, int N, const float *A, int lda, Epi epi, int nrt, const float *bias,
                                            const GxPreX *pre = nullptr) {
  const int lane = gx_lane(), j = lane & 15, g = lane >> 4;
  const int KB = (K + 31) >> 5, units = nrt * (N >> 5);
  gx_u4 ah[MAXKB], al[MAXKB];
  int rt_have = -1;
  gx_epi_pre(epi, 0, 0, 0);
  gx_epi_rotate(epi, 0);
  const gx_u4 *wl = reinterpret_cast<const gx_u4 *>(WX) + (size_t)lane * 4;
  gx_u4 wc0, wc1, wc2, wc3;
  if (pre != nullptr && pre->valid) { wc0 = pre->w0; wc1 = pre->w1; wc2 = pre->w2; wc3 = pre->w3; }
  else { wc0 = wl[0]; wc1 = wl[1]; wc2 = wl[2]; wc3 = wl[3]; }
  gx_u4 wn0 = wc0, wn1 = wc1, wn2 = wc2, wn3 = wc3;
  f32x2 bb = bias ? *reinterpret_cast<const f32x2 *>(bias + 2 * j) : f32x2{0.0f, 0.0f}, bn = bb;
  for (int u = 0; u < units; ++u) {
    const int rt = u % nrt, cg = u / nrt, n0 = cg << 5;
    const float *ap = A + (size_t)(16 * rt + j) * lda + 8 * g;
    const bool more = u + 1 < units;
    const int cgn = (u + 1) / nrt;
    if (more) { gx_epi_pre(epi, (u + 1) % nrt, cgn << 5, 0); if (bias) bn = *reinterpret_cast<const f32x2 *>(bias + (cgn << 5) + 2 * j); }
    f32x4 acc0 = {bb[0], bb[0], bb[0], bb[0]}, acc1 = {bb[1], bb[1], bb[1], bb[1]};
    for (int kc = 0; kc < KB; kc += MAXKB) {          // (one pass unless RESPLIT)
      if (RESPLIT || rt != rt_have) {
#pragma unroll
        for (int kb = 0; kb < MAXKB; ++kb)
          if (kc + kb < KB)
            gx_split8(*reinterpret_cast<const f32x4 *>(ap + 32 * (kc + kb)), *reinterpret_cast<const f32x4 *>(ap + 32 * (kc + kb) + 4), ah[kb], al[kb]);
        rt_have = rt;
      }
      // the two weight sets exchange roles by NAME from block to block (even blocks multiply out of wc and request into wn, odd ones the
      // other way round): a rotation wc = wn is eight 64-bit moves per block that need the requested data at once -- no block in flight
      auto next_into = [&](int kabs, gx_u4 &d0, gx_u4 &d1, gx_u4 &d2, gx_u4 &d3) {      // the stream's block behind (this unit, kabs)
        const bool last = kabs + 1 == KB;
        if (!last || more) {
          const gx_u4 *wq = wl + ((size_t)(last ? cgn : cg) * KB + (last ? 0 : kabs + 1)) * 256;
          d0 = wq[0]; d1 = wq[1]; d2 = wq[2]; d3 = wq[3];
        }
      };
#pragma unroll
      for (int kb = 0; kb < MAXKB; kb += 2) {
        if (kc + kb < KB) {
          next_into(kc + kb, wn0, wn1, wn2, wn3);
          __builtin_amdgcn_sched_barrier(0);
          acc0 = GX_MFMA_H(al[kb], wc0, acc0); acc1 = GX_MFMA_H(al[kb], wc1, acc1);      // the small cross terms first
          acc0 = GX_MFMA_H(ah[kb], wc2, acc0); acc1 = GX_MFMA_H(ah[kb], wc3, acc1);
          acc0 = GX_MFMA_H(ah[kb], wc0, acc0); acc1 = GX_MFMA_H(ah[kb], wc1, acc1);
        }
        if (kb + 1 < MAXKB && kc + kb + 1 < KB) {
          next_into(kc + kb + 1, wc0, wc1, wc2, wc3);
          __builtin_amdgcn_sched_barrier(0);
          acc0 = GX_MFMA_H(al[kb + 1], wn0, acc0); acc1 = GX_MFMA_H(al[kb + 1], wn1, acc1);
          acc0 = GX_MFMA_H(ah[kb + 1], wn2, acc0); acc1 = GX_MFMA_H(ah[kb + 1], wn3, acc1);
          acc0 = GX_MFMA_H(ah[kb + 1], wn0, acc0); acc1 = GX_MFMA_H(ah[kb + 1], wn1, acc1);
        }
      }
    }
    if (KB & 1) { wc0 = wn0; wc1 = wn1; wc2 = wn2; wc3 = wn3; }      // an odd number of blocks: the next unit's first block sits in the other set
    epi(rt, n0, acc0, acc1);
    if (more) gx_epi_rotate(epi, 0);
    bb = bn;
  }
}

// W [K][N] with row stride N
template <bool A_GLOBAL = false, bool FRAG = false, class Epi>
__device__ __forceinline__ void gx_dense(const float *__restrict__ W, int K, int N, const float *A, int lda, Epi epi, int nrt = 2,
                                         const float *bias = nullptr, const GxPre *pre = nullptr, int u0 = -1, int ustride = GX_WAVES) {
  gx_dense_ld<A_GLOBAL, FRAG>(W, N, K, N, A, lda, epi, nrt, bias, pre, u0, ustride);
}

// Epilogue helpers -------------------------------------------------------------------------------------------------------------
// y = act(acc + b) -> Y (LDS [32][ldy]); LEAKY: LeakyReLU(0.2), else linear
// (bias == NULL: the engine started the accumulators from the bias)
// the lane's own part of a store address, (4 g) ldy + 2 j: formed once per layer walk by the row-tile-per-wave kernels (GxStore::off; the
// opaque lane copy otherwise makes every epilogue call re-derive it -- two quarter-rate integer multiplies per unit)
__device__ __forceinline__ int gx_store_off(int ldy) { const int lane = gx_lane(); return (4 * (lane >> 4)) * ldy + 2 * (lane & 15); }
template <bool LEAKY>
struct GxStore {
  float *Y; int ldy; const float *bias;
  int off = -1;               // gx_store_off(ldy), or -1: derive it per call
  __device__ __forceinline__ void operator()(int rt, int n0, const f32x4 &a0, const f32x4 &a1) const {
    const int lane = gx_lane(), j = lane & 15;
    const f32x2 bb = bias ? *reinterpret_cast<const f32x2 *>(bias + n0 + 2 * j) : f32x2{0.0f, 0.0f};
    float *yp = Y + (off >= 0 ? off : gx_store_off(ldy)) + (16 * rt) * ldy + n0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float y0 = bias ? a0[r] + bb[0] : a0[r], y1 = bias ? a1[r] + bb[1] : a1[r];      // (x + 0.0f is not folded: -0.0f)
      if (LEAKY) { y0 = lrelu(y0); y1 = lrelu(y1); }
      f32x2 o = {y0, y1};
      *reinterpret_cast<f32x2 *>(yp + r * ldy) = o;
    }
  }
};

// The same, and a copy of the activation rows to a global workspace G [rows][ldg] (fit: what the weight-gradient GEMM reads)
template <bool LEAKY>
struct GxStoreWs {
  float *Y; int ldy; const float *bias; float *G; int ldg;
  __device__ __forceinline__ void operator()(int rt, int n0, const f32x4 &a0, const f32x4 &a1) const {
    const int lane = gx_lane(), j = lane & 15, g = lane >> 4;
    const f32x2 bb = bias ? *reinterpret_cast<const f32x2 *>(bias + n0 + 2 * j) : f32x2{0.0f, 0.0f};      // (NULL: already in the accumulators)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float y0 = bias ? a0[r] + bb[0] : a0[r], y1 = bias ? a1[r] + bb[1] : a1[r];
      if (LEAKY) { y0 = lrelu(y0); y1 = lrelu(y1); }
      f32x2 o = {y0, y1};
      const int row = 16 * rt + 4 * g + r;
      if (Y) *reinterpret_cast<f32x2 *>(Y + (size_t)row * ldy + n0 + 2 * j) = o;
      *reinterpret_cast<f32x2 *>(G + (size_t)row * ldg + n0 + 2 * j) = o;
    }
  }
};

// Backward through a LeakyReLU layer: d(pre-activation) = dX * (h > 0 ? 1 : 0.2) with h the stored post-activation of that layer
// (sign(h) = sign(pre-activation)); written to LDS (next A operand) and to the global workspace D (weight gradients), H global.
struct GxBackStore {
  float *Y; int ldy; const float *H; int ldh; float *D; int ldd;
  __device__ __forceinline__ void operator()(int rt, int n0, const f32x4 &a0, const f32x4 &a1) const {
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * rt + 4 * g + r;
      const f32x2 hh = *reinterpret_cast<const f32x2 *>(H + (size_t)row * ldh + n0 + 2 * j);
      f32x2 o = {a0[r] * (hh[0] > 0.0f ? 1.0f : BGM_LEAK), a1[r] * (hh[1] > 0.0f ? 1.0f : BGM_LEAK)};
      *reinterpret_cast<f32x2 *>(Y + (size_t)row * ldy + n0 + 2 * j) = o;
      if (D) *reinterpret_cast<f32x2 *>(D + (size_t)row * ldd + n0 + 2 * j) = o;
    }
  }
};

// plain store of the accumulators (gradient with respect to a network input)
struct GxRawStore {
  float *Y; int ldy;
  __device__ __forceinline__ void operator()(int rt, int n0, const f32x4 &a0, const f32x4 &a1) const {
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x2 o = {a0[r], a1[r]};
      *reinterpret_cast<f32x2 *>(Y + (size_t)(16 * rt + 4 * g + r) * ldy + n0 + 2 * j) = o;
    }
  }
};

// Hidden layers l = l_begin .. l_end - 1 of `net` (LeakyReLU), ping-ponging between two LDS buffers; the input is in `cur`.
// Returns with the last output in the returned buffer (a barrier has been passed after its last store).  pre: in = layer l_begin's first
// block (gx_prefetch; or valid = 0), out = layer l_end's when the net has one -- every layer's first weights and biases are requested
// before the previous layer's products and arrive under them and the barrier.
__device__ __forceinline__ float *gx_hidden(const GxNet &net, const float *pack, int l_begin, int l_end, float *cur, float *oth, int ld, GxPre &pre,
                                            int nrt = 2, int nrt_after = 2) {
  const int soff = gx_store_off(ld);
  for (int l = l_begin; l < l_end; ++l) {
    GxPre nx;
    nx.valid = 0;
    if (l + 1 < net.L) nx = gx_prefetch(pack + net.w[l + 1], net.pad[l + 2], net.pad[l + 2], pack + net.b[l + 1], l + 1 < l_end ? nrt : nrt_after);
    gx_dense(pack + net.w[l], net.pad[l], net.pad[l + 1], cur, ld, GxStore<true>{oth, ld, nullptr, soff}, nrt, pack + net.b[l], &pre);
    __syncthreads();
    pre = nx;
    float *t = cur; cur = oth; oth = t;
  }
  return cur;
}

// Sum of v over the 16 lanes j of a lane group (all lanes get the total).
__device__ __forceinline__ float gx_sum_j(float v) {
  v += __shfl_xor(v, 1, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 8, 16);
  return v;
}
