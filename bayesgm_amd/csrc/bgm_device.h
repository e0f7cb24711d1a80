// bgm_device.h -- gfx950 device building blocks shared by the hot-path kernels.
//
// Execution layout ("swapped" MFMA orientation)
// ---------------------------------------------
// Every dense layer  H_out = act(H_in W + b)  of the reference's MLPs
// (networks/base.py:30-51) is evaluated as  H_out^T = W^T H_in^T  with
// v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles/SIMD, D = A[16x4] B[4x16] + C):
//     M (i) = output feature within a 16-wide tile      -> A operand = weights
//     N (j) = chain / observation row (16 per MFMA)      -> B operand = activations
//     K     = input feature
// Lane l = (j = l & 15, g = l >> 4).  Result register r of tile t holds output
// feature 16 t + 4 g + r of row j; since the contraction order is free, K-step
// (t, r) of the NEXT layer contracts over features {16 t + 4 g + r : g = 0..3},
// i.e. the accumulator registers of one layer ARE the B operands of the next:
// no transpose, no LDS round trip, no cross-lane traffic between layers.  Row
// reductions (sum over features of one chain) are per-lane sums plus two
// cross-lane adds over g.
//
// Weights sit in LDS in fragment order: a layer with K_ROWS = 16*KT packed input
// rows and NT output tiles is stored as consecutive tile groups of GS = 4/2/1
// tiles, each [K_ROWS][16 (j)][GS] floats, so that one ds_read_b128/b64/b32 per
// K-step yields the A fragments of GS tiles (element (rho, j, u) = W[rho][16(t0+u)+j]).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The LDS-resident weights never change inside a kernel, so LLVM's LICM hoists
// every weight read out of the iteration loops and spills it.  A compiler-only
// memory clobber at the top of each iteration keeps the reads where they are used.
#define BGM_NO_HOIST() asm volatile("" ::: "memory")
// The same LICM also hoists every loop-invariant LDS *address* (lane offset + layer offset) out of the
// iteration loops and keeps dozens of them live.  Laundering the lane indices through an empty asm at
// the entry of a layer makes the addresses be formed where they are used.
#define BGM_OPAQUE2(a, b) asm volatile("" : "+v"(a), "+v"(b))

#define BGM_LEAK 0.2f
#define BGM_EPS 1e-6f

// ---------------------------------------------------------------- RNG --------
// Philox4x32-10 (Salmon et al. 2011); spec restated in oracle/rng.py.
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

enum { TAG_INIT = 0, TAG_PROP = 1, TAG_ACC = 2, TAG_YNOISE = 3, TAG_MOM = 4, TAG_HACC = 5, TAG_XNOISE = 6 };

__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(PHILOX_M0, c0), lo0 = PHILOX_M0 * c0;
    const uint32_t hi1 = __umulhi(PHILOX_M1, c2), lo1 = PHILOX_M1 * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += PHILOX_W0;
    k1 += PHILOX_W1;
  }
  return make_uint4(c0, c1, c2, c3);
}

__device__ __forceinline__ float u01_open(uint32_t x) {  // (0,1)
  return ((float)(x >> 8) + 0.5f) * 5.9604644775390625e-8f;
}
__device__ __forceinline__ float u01_half(uint32_t x) {  // [0,1)
  return (float)(x >> 8) * 5.9604644775390625e-8f;
}

// 4 words -> 4 standard normals: r = sqrt(-2 ln u1), (r cos 2 pi u2, r sin 2 pi u2).
// v_log_f32 is log2, v_sin/v_cos take revolutions.
__device__ __forceinline__ f32x4 box_muller4(uint4 w) {
  f32x4 o;
  {
    const float u1 = u01_open(w.x), u2 = u01_half(w.y);
    const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    o[0] = r * __builtin_amdgcn_cosf(u2);
    o[1] = r * __builtin_amdgcn_sinf(u2);
  }
  {
    const float u1 = u01_open(w.z), u2 = u01_half(w.w);
    const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    o[2] = r * __builtin_amdgcn_cosf(u2);
    o[3] = r * __builtin_amdgcn_sinf(u2);
  }
  return o;
}

// ------------------------------------------------------------- math ----------
// max(a,b) as v_med3_f32(a, b, +inf): a single VOP3 without the NaN-canonicalising extra v_max
// that fmaxf() costs on MFMA outputs.  (An inline-asm v_max_f32 is NOT an option here: hipcc pads
// no MFMA->VALU hazard wait states inside asm statements, so it would read stale accumulators.)
__device__ __forceinline__ float vmax(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
__device__ __forceinline__ float lrelu(float x) { return vmax(x, BGM_LEAK * x); }
// natural log / exp on the hardware transcendental units (v_log_f32 = log2, v_exp_f32 = exp2;
// ~1 ulp each), used where the argument is O(1) and the result enters a sum of O(1e2) terms.
__device__ __forceinline__ float fast_log(float x) { return 0.6931471805599453f * __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(1.4426950408889634f * x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// tf.nn.softplus(x) = log(1 + exp(x)) = max(x,0) + log(1 + exp(-|x|)); for e = exp(-|x|) < 2^-12
// log(1+e) = e - e^2/2 + O(e^3) is used so that tiny tails keep their relative accuracy.
__device__ __forceinline__ float softplus_f(float x) {
  const float e = fast_exp(-fabsf(x));
  const float l = (e < 2.44140625e-4f) ? e * (1.0f - 0.5f * e) : fast_log(1.0f + e);
  return vmax(x, 0.0f) + l;
}

// accurate (libm) forms, used where a value is reported or differentiated in the fit path
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float softplus_acc(float x) { return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x))); }

// sum over the four lane groups g (lanes j, j+16, j+32, j+48); result in all.
__device__ __forceinline__ float sum_over_g(float x) {
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}
// sum over the 16 rows j of a lane group with DPP row shifts (no LDS traffic);
// the total lands in lane j = 15 of each group.
__device__ __forceinline__ float sum_over_j_to_lane15(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xF, 0xF, true));  // row_shr:1
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xF, 0xF, true));  // row_shr:2
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xF, 0xF, true));  // row_shr:4
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xF, 0xF, true));  // row_shr:8
  return x;
}

// ------------------------------------------------------- MFMA dense layer ----
#define BGM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int GS>
struct AFrag;
template <>
struct AFrag<4> {
  f32x4 v;
  __device__ __forceinline__ void load(const float *p) { v = *reinterpret_cast<const f32x4 *>(p); }
  __device__ __forceinline__ float get(int u) const { return v[u]; }
};
template <>
struct AFrag<2> {
  f32x2 v;
  __device__ __forceinline__ void load(const float *p) { v = *reinterpret_cast<const f32x2 *>(p); }
  __device__ __forceinline__ float get(int u) const { return v[u]; }
};
template <>
struct AFrag<1> {
  float v;
  __device__ __forceinline__ void load(const float *p) { v = *p; }
  __device__ __forceinline__ float get(int) const { return v; }
};

#ifndef BGM_MAX_GROUP
#define BGM_MAX_GROUP 4
#endif
__host__ __device__ constexpr int group_size(int tiles_left) {
  return (tiles_left >= 4 && BGM_MAX_GROUP >= 4) ? 4 : (tiles_left >= 2 ? 2 : 1);
}

// One tile group [T0, T0+GS) of a layer.  KT input tiles, the last of which
// uses KSL (1..4) K-steps.  `wl` points at the layer's packed weights in LDS,
// `lane_off` = (64 g + j).
template <int T0, int KT, int KSL, int NT, int R>
__device__ __forceinline__ void dense_groups(const float *wl, int lane_off, const f32x4 (&in)[R][KT],
                                             f32x4 (&acc)[R][NT]) {
  if constexpr (T0 < NT) {
    constexpr int GS = group_size(NT - T0);
    constexpr int K_ROWS = 16 * KT;
    constexpr int NKS = 4 * (KT - 1) + KSL;  // K-steps of this layer
    const float *base = wl + K_ROWS * 16 * T0 + lane_off * GS;
    // software pipeline in SOURCE order: the A fragment of K-step s+1 is requested before the MFMAs of
    // step s.  (hipcc sinks the load back to its first use; pinning the order with sched_barrier(0) or
    // inline-asm ds_read + counted lgkmcnt gives the ideal stream but ~300 spilled VGPRs and 68 vs 113 TF --
    // measured, see DESIGN.md -- so the load placement is left to the compiler and covered by the
    // second wave of the SIMD.)
    AFrag<GS> a_cur, a_nxt;
    a_cur.load(base);
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const int t = s >> 2, r = s & 3;
      if (s + 1 < NKS) a_nxt.load(base + (16 * ((s + 1) >> 2) + ((s + 1) & 3)) * 16 * GS);
#pragma unroll
      for (int u = 0; u < GS; ++u) {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) acc[rr][T0 + u] = BGM_MFMA(a_cur.get(u), in[rr][t][r], acc[rr][T0 + u]);
      }
      a_cur = a_nxt;
    }
    dense_groups<T0 + GS, KT, KSL, NT, R>(wl, lane_off, in, acc);
  }
}

// acc = bias (feature 16 t + 4 g + r) for every row group
template <int NT, int R>
__device__ __forceinline__ void bias_init(const float *bl, int g, f32x4 (&acc)[R][NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x4 b = *reinterpret_cast<const f32x4 *>(bl + 16 * t + 4 * g);
#pragma unroll
    for (int rr = 0; rr < R; ++rr) acc[rr][t] = b;
  }
}

template <int KT, int KSL, int NT, int R>
__device__ __forceinline__ void dense(const float *wl, const float *bl, int lane_off, int g,
                                      const f32x4 (&in)[R][KT], f32x4 (&acc)[R][NT]) {
  bias_init<NT, R>(bl, g, acc);
  dense_groups<0, KT, KSL, NT, R>(wl, lane_off, in, acc);
}

template <int NT, int R>
__device__ __forceinline__ void lrelu_inplace(f32x4 (&a)[R][NT]) {
#pragma unroll
  for (int rr = 0; rr < R; ++rr)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) a[rr][t][r] = lrelu(a[rr][t][r]);
}

// copy a packed weight blob (global) into LDS, all threads of the block
__device__ __forceinline__ void lds_fill(float *lds, const float *blob, int total_floats) {
  const f32x4 *src = reinterpret_cast<const f32x4 *>(blob);
  f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
  for (int i = threadIdx.x; i < total_floats / 4; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
}
