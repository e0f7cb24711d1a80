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
// LeakyReLU(0.2) up to the positive factor 0.6 in ONE instruction: x + (2/3)|x| = LeakyReLU(x) / 0.6 (v_fma_f32 with the
// |.| input modifier; max(x, 0.2 x) costs a multiply and a max).  The sampling kernels of CausalBGM use it with the
// weights of every layer that consumes such an activation scaled by 0.6 in their copy of the packed blob
// (causal_scale_blob_kernel), so that W'(h / 0.6) = W h: the result differs from the max form by fp32 rounding only.
#define BGM_LRS 0.6666666865348816f
#define BGM_LRS_W 0.6f
__device__ __forceinline__ float lrelu_s(float x) { return fmaf(fabsf(x), BGM_LRS, x); }
// Same value, pinned in a register by an empty asm: with the plain form at every site of causal_effects hipcc 7.2 dies with
// "Illegal instruction detected: Operand has incorrect register class ... $src_shared_base" (the mis-selected flat <-> LDS
// null check also noted at lds_byte_addr); pinning the value at the one hot site moves the code out of that pattern.
__device__ __forceinline__ float lrelu_s_pinned(float x) { float r = fmaf(fabsf(x), BGM_LRS, x); asm volatile("" : "+v"(r)); return r; }
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
#ifndef BGM_PREFETCH_ALL_MAX
#define BGM_PREFETCH_ALL_MAX 32   // fragment registers a small layer may request up front (0 disables)
#endif
__host__ __device__ constexpr int group_size(int tiles_left) {
  return (tiles_left >= 4 && BGM_MAX_GROUP >= 4) ? 4 : (tiles_left >= 2 ? 2 : 1);
}

// ---------------------------------------------------------------------------------------------
// Hand-scheduled tile group: acc[0..3] += W[64 K-rows x 4 tiles] (packed group, GS = 4) * in.
// hipcc sinks every LDS fragment load to its first use (ds_read; s_waitcnt lgkmcnt(0); 4 x MFMA), which
// exposes the LDS latency once per K-step; pinning the order from C++ made the register allocator
// spill.  Here the 16 K-steps are one asm block with its own three A-fragment buffers (fixed
// v244-v255, declared clobbered): fragments are requested TWO steps ahead and waited for with
// counted lgkmcnt, so a step's four MFMAs issue back to back.  The block drains every older LGKM
// operation first (SMEM returns out of order, which would break counted waits) and ends with the
// wait states the MFMA -> VALU read hazard needs (the compiler does not see the MFMAs).
// `lds_addr` = LDS byte address of this lane's fragment of K-step 0.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void dense_group4_k64_asm(unsigned lds_addr, const f32x4 (&in)[4], f32x4 &a0, f32x4 &a1, f32x4 &a2,
                                                     f32x4 &a3) {
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n"
      "ds_read_b128 v[244:247], %20 offset:0\n"
      "ds_read_b128 v[248:251], %20 offset:256\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v244, %4, %0\n"
      "ds_read_b128 v[252:255], %20 offset:512\n"
      "v_mfma_f32_16x16x4_f32 %1, v245, %4, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v246, %4, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v247, %4, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v248, %5, %0\n"
      "ds_read_b128 v[244:247], %20 offset:768\n"
      "v_mfma_f32_16x16x4_f32 %1, v249, %5, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v250, %5, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v251, %5, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v252, %6, %0\n"
      "ds_read_b128 v[248:251], %20 offset:4096\n"
      "v_mfma_f32_16x16x4_f32 %1, v253, %6, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v254, %6, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v255, %6, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v244, %7, %0\n"
      "ds_read_b128 v[252:255], %20 offset:4352\n"
      "v_mfma_f32_16x16x4_f32 %1, v245, %7, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v246, %7, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v247, %7, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v248, %8, %0\n"
      "ds_read_b128 v[244:247], %20 offset:4608\n"
      "v_mfma_f32_16x16x4_f32 %1, v249, %8, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v250, %8, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v251, %8, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v252, %9, %0\n"
      "ds_read_b128 v[248:251], %20 offset:4864\n"
      "v_mfma_f32_16x16x4_f32 %1, v253, %9, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v254, %9, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v255, %9, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v244, %10, %0\n"
      "ds_read_b128 v[252:255], %20 offset:8192\n"
      "v_mfma_f32_16x16x4_f32 %1, v245, %10, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v246, %10, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v247, %10, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v248, %11, %0\n"
      "ds_read_b128 v[244:247], %20 offset:8448\n"
      "v_mfma_f32_16x16x4_f32 %1, v249, %11, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v250, %11, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v251, %11, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v252, %12, %0\n"
      "ds_read_b128 v[248:251], %20 offset:8704\n"
      "v_mfma_f32_16x16x4_f32 %1, v253, %12, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v254, %12, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v255, %12, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v244, %13, %0\n"
      "ds_read_b128 v[252:255], %20 offset:8960\n"
      "v_mfma_f32_16x16x4_f32 %1, v245, %13, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v246, %13, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v247, %13, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v248, %14, %0\n"
      "ds_read_b128 v[244:247], %20 offset:12288\n"
      "v_mfma_f32_16x16x4_f32 %1, v249, %14, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v250, %14, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v251, %14, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v252, %15, %0\n"
      "ds_read_b128 v[248:251], %20 offset:12544\n"
      "v_mfma_f32_16x16x4_f32 %1, v253, %15, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v254, %15, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v255, %15, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v244, %16, %0\n"
      "ds_read_b128 v[252:255], %20 offset:12800\n"
      "v_mfma_f32_16x16x4_f32 %1, v245, %16, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v246, %16, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v247, %16, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v248, %17, %0\n"
      "ds_read_b128 v[244:247], %20 offset:13056\n"
      "v_mfma_f32_16x16x4_f32 %1, v249, %17, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v250, %17, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v251, %17, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v252, %18, %0\n"
      "v_mfma_f32_16x16x4_f32 %1, v253, %18, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v254, %18, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v255, %18, %3\n"
      "s_waitcnt lgkmcnt(0)\n"
      "v_mfma_f32_16x16x4_f32 %0, v244, %19, %0\n"
      "v_mfma_f32_16x16x4_f32 %1, v245, %19, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v246, %19, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v247, %19, %3\n"
      "s_nop 15\n"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
      : "v"(in[0][0]), "v"(in[0][1]), "v"(in[0][2]), "v"(in[0][3]), "v"(in[1][0]), "v"(in[1][1]), "v"(in[1][2]), "v"(in[1][3]),
        "v"(in[2][0]), "v"(in[2][1]), "v"(in[2][2]), "v"(in[2][3]), "v"(in[3][0]), "v"(in[3][1]), "v"(in[3][2]), "v"(in[3][3]),
        "v"(lds_addr)
      : "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
}
// Same block with the accumulators STARTING from separate registers c0..c3 (the C operand of the first K-step): the
// last layer of g starts from (bias - v), kept in registers for the whole kernel, so its result is (mu - v) and the
// likelihood epilogue needs neither a bias load nor a subtraction.
__device__ __forceinline__ void dense_group4_k64_asm_c(unsigned lds_addr, const f32x4 (&in)[4], const f32x4 &c0, const f32x4 &c1, const f32x4 &c2,
                                                       const f32x4 &c3, f32x4 &a0, f32x4 &a1, f32x4 &a2, f32x4 &a3) {
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n"
      "ds_read_b128 v[244:247], %20 offset:0\n"
      "ds_read_b128 v[248:251], %20 offset:256\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v244, %4, %21\n"
      "ds_read_b128 v[252:255], %20 offset:512\n"
      "v_mfma_f32_16x16x4_f32 %1, v245, %4, %22\n"
      "v_mfma_f32_16x16x4_f32 %2, v246, %4, %23\n"
      "v_mfma_f32_16x16x4_f32 %3, v247, %4, %24\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v248, %5, %0\n"
      "ds_read_b128 v[244:247], %20 offset:768\n"
      "v_mfma_f32_16x16x4_f32 %1, v249, %5, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v250, %5, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v251, %5, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v252, %6, %0\n"
      "ds_read_b128 v[248:251], %20 offset:4096\n"
      "v_mfma_f32_16x16x4_f32 %1, v253, %6, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v254, %6, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v255, %6, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v244, %7, %0\n"
      "ds_read_b128 v[252:255], %20 offset:4352\n"
      "v_mfma_f32_16x16x4_f32 %1, v245, %7, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v246, %7, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v247, %7, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v248, %8, %0\n"
      "ds_read_b128 v[244:247], %20 offset:4608\n"
      "v_mfma_f32_16x16x4_f32 %1, v249, %8, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v250, %8, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v251, %8, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v252, %9, %0\n"
      "ds_read_b128 v[248:251], %20 offset:4864\n"
      "v_mfma_f32_16x16x4_f32 %1, v253, %9, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v254, %9, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v255, %9, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v244, %10, %0\n"
      "ds_read_b128 v[252:255], %20 offset:8192\n"
      "v_mfma_f32_16x16x4_f32 %1, v245, %10, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v246, %10, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v247, %10, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v248, %11, %0\n"
      "ds_read_b128 v[244:247], %20 offset:8448\n"
      "v_mfma_f32_16x16x4_f32 %1, v249, %11, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v250, %11, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v251, %11, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v252, %12, %0\n"
      "ds_read_b128 v[248:251], %20 offset:8704\n"
      "v_mfma_f32_16x16x4_f32 %1, v253, %12, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v254, %12, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v255, %12, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v244, %13, %0\n"
      "ds_read_b128 v[252:255], %20 offset:8960\n"
      "v_mfma_f32_16x16x4_f32 %1, v245, %13, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v246, %13, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v247, %13, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v248, %14, %0\n"
      "ds_read_b128 v[244:247], %20 offset:12288\n"
      "v_mfma_f32_16x16x4_f32 %1, v249, %14, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v250, %14, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v251, %14, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v252, %15, %0\n"
      "ds_read_b128 v[248:251], %20 offset:12544\n"
      "v_mfma_f32_16x16x4_f32 %1, v253, %15, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v254, %15, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v255, %15, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v244, %16, %0\n"
      "ds_read_b128 v[252:255], %20 offset:12800\n"
      "v_mfma_f32_16x16x4_f32 %1, v245, %16, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v246, %16, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v247, %16, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v248, %17, %0\n"
      "ds_read_b128 v[244:247], %20 offset:13056\n"
      "v_mfma_f32_16x16x4_f32 %1, v249, %17, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v250, %17, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v251, %17, %3\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 %0, v252, %18, %0\n"
      "v_mfma_f32_16x16x4_f32 %1, v253, %18, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v254, %18, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v255, %18, %3\n"
      "s_waitcnt lgkmcnt(0)\n"
      "v_mfma_f32_16x16x4_f32 %0, v244, %19, %0\n"
      "v_mfma_f32_16x16x4_f32 %1, v245, %19, %1\n"
      "v_mfma_f32_16x16x4_f32 %2, v246, %19, %2\n"
      "v_mfma_f32_16x16x4_f32 %3, v247, %19, %3\n"
      "s_nop 15\n"
      : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
      : "v"(in[0][0]), "v"(in[0][1]), "v"(in[0][2]), "v"(in[0][3]), "v"(in[1][0]), "v"(in[1][1]), "v"(in[1][2]), "v"(in[1][3]),
        "v"(in[2][0]), "v"(in[2][1]), "v"(in[2][2]), "v"(in[2][3]), "v"(in[3][0]), "v"(in[3][1]), "v"(in[3][2]), "v"(in[3][3]),
        "v"(lds_addr), "v"(c0), "v"(c1), "v"(c2), "v"(c3)
      : "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
}
__device__ __forceinline__ unsigned lds_byte_addr(const float *p) {
  // LDS byte address of an element of the kernel's dynamic shared array, as (offset of that array) + (p - array).
  // The pointer difference cancels the flat aperture, so no flat <-> LDS cast of `p` is ever materialised: hipcc 7.2
  // mis-selects those casts in some inlining contexts ("Illegal instruction detected ... $src_shared_base"), both for
  // an explicit address_space(3) cast and for truncating the flat address.
  extern __shared__ __attribute__((aligned(16))) float bgm_dynamic_lds[];
  return (unsigned)__builtin_amdgcn_groupstaticsize() + 4u * (unsigned)(p - bgm_dynamic_lds);
}

// ---------------------------------------------------------------------------------------------
// The four hidden->hidden layers of the g net (64 x 64 each) as ONE hand-scheduled block.
// Between two layers the compiler-scheduled path pays bias loads + the first A fragments + the MFMA drain + a
// 32-instruction LeakyReLU, ~580 cycles per layer with the matrix pipe empty (measured).  Here the activation sets
// live in fixed registers (P = v208-v223, Q = v224-v239) and alternate as B operands / accumulators; a layer's
// accumulator tuples are loaded with the bias as soon as the previous layer has consumed them as inputs, the A
// fragments stream two K-steps ahead ACROSS layer boundaries, and the (scaled, one-instruction) LeakyReLU lrelu_s is applied
// to one input element per K-step, just in time (v243 holds the constant 2/3).  Software-managed hazards: counted lgkmcnt per LDS load, s_nop between the last MFMA of a
// layer and the first VALU read of its result.
//   w_addr: LDS byte address of this lane's fragment of layer 0, K-step 0 (layers are 16 KiB apart);
//   b_addr: LDS byte address of bias feature 4g of layer 0 (layers 256 B apart).
//   p: in = activated input (lrelu_s), out = RAW output of the 4th layer (caller applies lrelu_s);  q: scratch set.
//   The weights of all four layers must carry the factor 0.6 (sampling copy of the blob).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void dense_hidden4_asm(unsigned w_addr, unsigned b_addr, f32x4 (&p)[4], f32x4 (&q)[4]) {
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n"
      "v_mov_b32 v243, 0x3f2aaaab\n"
      "ds_read_b128 v[224:227], %9 offset:0\n"
      "ds_read_b128 v[228:231], %9 offset:64\n"
      "ds_read_b128 v[232:235], %9 offset:128\n"
      "ds_read_b128 v[236:239], %9 offset:192\n"
      "ds_read_b128 v[244:247], %8 offset:0\n"
      "ds_read_b128 v[248:251], %8 offset:256\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v244, v208, v[224:227]\n"
      "ds_read_b128 v[252:255], %8 offset:512\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v245, v208, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v246, v208, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v247, v208, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v248, v209, v[224:227]\n"
      "ds_read_b128 v[244:247], %8 offset:768\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v249, v209, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v250, v209, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v251, v209, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v252, v210, v[224:227]\n"
      "ds_read_b128 v[248:251], %8 offset:4096\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v253, v210, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v254, v210, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v255, v210, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v244, v211, v[224:227]\n"
      "ds_read_b128 v[252:255], %8 offset:4352\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v245, v211, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v246, v211, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v247, v211, v[236:239]\n"
      "ds_read_b128 v[208:211], %9 offset:256\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v248, v212, v[224:227]\n"
      "ds_read_b128 v[244:247], %8 offset:4608\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v249, v212, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v250, v212, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v251, v212, v[236:239]\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v252, v213, v[224:227]\n"
      "ds_read_b128 v[248:251], %8 offset:4864\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v253, v213, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v254, v213, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v255, v213, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v244, v214, v[224:227]\n"
      "ds_read_b128 v[252:255], %8 offset:8192\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v245, v214, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v246, v214, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v247, v214, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v248, v215, v[224:227]\n"
      "ds_read_b128 v[244:247], %8 offset:8448\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v249, v215, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v250, v215, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v251, v215, v[236:239]\n"
      "ds_read_b128 v[212:215], %9 offset:320\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v252, v216, v[224:227]\n"
      "ds_read_b128 v[248:251], %8 offset:8704\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v253, v216, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v254, v216, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v255, v216, v[236:239]\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v244, v217, v[224:227]\n"
      "ds_read_b128 v[252:255], %8 offset:8960\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v245, v217, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v246, v217, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v247, v217, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v248, v218, v[224:227]\n"
      "ds_read_b128 v[244:247], %8 offset:12288\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v249, v218, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v250, v218, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v251, v218, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v252, v219, v[224:227]\n"
      "ds_read_b128 v[248:251], %8 offset:12544\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v253, v219, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v254, v219, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v255, v219, v[236:239]\n"
      "ds_read_b128 v[216:219], %9 offset:384\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v244, v220, v[224:227]\n"
      "ds_read_b128 v[252:255], %8 offset:12800\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v245, v220, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v246, v220, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v247, v220, v[236:239]\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v248, v221, v[224:227]\n"
      "ds_read_b128 v[244:247], %8 offset:13056\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v249, v221, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v250, v221, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v251, v221, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v252, v222, v[224:227]\n"
      "ds_read_b128 v[248:251], %8 offset:16384\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v253, v222, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v254, v222, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v255, v222, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v244, v223, v[224:227]\n"
      "ds_read_b128 v[252:255], %8 offset:16640\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v245, v223, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v246, v223, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v247, v223, v[236:239]\n"
      "ds_read_b128 v[220:223], %9 offset:448\n"
      "s_nop 15\n"
      "v_fma_f32 v224, |v224|, v243, v224\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v225, |v225|, v243, v225\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v248, v224, v[208:211]\n"
      "ds_read_b128 v[244:247], %8 offset:16896\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v249, v224, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v250, v224, v[216:219]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v251, v224, v[220:223]\n"
      "v_fma_f32 v226, |v226|, v243, v226\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v252, v225, v[208:211]\n"
      "ds_read_b128 v[248:251], %8 offset:17152\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v253, v225, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v254, v225, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v255, v225, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v227, |v227|, v243, v227\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v244, v226, v[208:211]\n"
      "ds_read_b128 v[252:255], %8 offset:20480\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v245, v226, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v246, v226, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v247, v226, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v228, |v228|, v243, v228\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v248, v227, v[208:211]\n"
      "ds_read_b128 v[244:247], %8 offset:20736\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v249, v227, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v250, v227, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v251, v227, v[220:223]\n"
      "ds_read_b128 v[224:227], %9 offset:512\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v229, |v229|, v243, v229\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v252, v228, v[208:211]\n"
      "ds_read_b128 v[248:251], %8 offset:20992\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v253, v228, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v254, v228, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v255, v228, v[220:223]\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v230, |v230|, v243, v230\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v244, v229, v[208:211]\n"
      "ds_read_b128 v[252:255], %8 offset:21248\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v245, v229, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v246, v229, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v247, v229, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v231, |v231|, v243, v231\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v248, v230, v[208:211]\n"
      "ds_read_b128 v[244:247], %8 offset:24576\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v249, v230, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v250, v230, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v251, v230, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v232, |v232|, v243, v232\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v252, v231, v[208:211]\n"
      "ds_read_b128 v[248:251], %8 offset:24832\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v253, v231, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v254, v231, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v255, v231, v[220:223]\n"
      "ds_read_b128 v[228:231], %9 offset:576\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v233, |v233|, v243, v233\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v244, v232, v[208:211]\n"
      "ds_read_b128 v[252:255], %8 offset:25088\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v245, v232, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v246, v232, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v247, v232, v[220:223]\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v234, |v234|, v243, v234\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v248, v233, v[208:211]\n"
      "ds_read_b128 v[244:247], %8 offset:25344\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v249, v233, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v250, v233, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v251, v233, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v235, |v235|, v243, v235\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v252, v234, v[208:211]\n"
      "ds_read_b128 v[248:251], %8 offset:28672\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v253, v234, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v254, v234, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v255, v234, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v236, |v236|, v243, v236\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v244, v235, v[208:211]\n"
      "ds_read_b128 v[252:255], %8 offset:28928\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v245, v235, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v246, v235, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v247, v235, v[220:223]\n"
      "ds_read_b128 v[232:235], %9 offset:640\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v237, |v237|, v243, v237\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v248, v236, v[208:211]\n"
      "ds_read_b128 v[244:247], %8 offset:29184\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v249, v236, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v250, v236, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v251, v236, v[220:223]\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v238, |v238|, v243, v238\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v252, v237, v[208:211]\n"
      "ds_read_b128 v[248:251], %8 offset:29440\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v253, v237, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v254, v237, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v255, v237, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v239, |v239|, v243, v239\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v244, v238, v[208:211]\n"
      "ds_read_b128 v[252:255], %8 offset:32768\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v245, v238, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v246, v238, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v247, v238, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v248, v239, v[208:211]\n"
      "ds_read_b128 v[244:247], %8 offset:33024\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v249, v239, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v250, v239, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v251, v239, v[220:223]\n"
      "ds_read_b128 v[236:239], %9 offset:704\n"
      "s_nop 15\n"
      "v_fma_f32 v208, |v208|, v243, v208\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v209, |v209|, v243, v209\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v252, v208, v[224:227]\n"
      "ds_read_b128 v[248:251], %8 offset:33280\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v253, v208, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v254, v208, v[232:235]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v255, v208, v[236:239]\n"
      "v_fma_f32 v210, |v210|, v243, v210\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v244, v209, v[224:227]\n"
      "ds_read_b128 v[252:255], %8 offset:33536\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v245, v209, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v246, v209, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v247, v209, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v211, |v211|, v243, v211\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v248, v210, v[224:227]\n"
      "ds_read_b128 v[244:247], %8 offset:36864\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v249, v210, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v250, v210, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v251, v210, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v212, |v212|, v243, v212\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v252, v211, v[224:227]\n"
      "ds_read_b128 v[248:251], %8 offset:37120\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v253, v211, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v254, v211, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v255, v211, v[236:239]\n"
      "ds_read_b128 v[208:211], %9 offset:768\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v213, |v213|, v243, v213\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v244, v212, v[224:227]\n"
      "ds_read_b128 v[252:255], %8 offset:37376\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v245, v212, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v246, v212, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v247, v212, v[236:239]\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v214, |v214|, v243, v214\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v248, v213, v[224:227]\n"
      "ds_read_b128 v[244:247], %8 offset:37632\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v249, v213, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v250, v213, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v251, v213, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v215, |v215|, v243, v215\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v252, v214, v[224:227]\n"
      "ds_read_b128 v[248:251], %8 offset:40960\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v253, v214, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v254, v214, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v255, v214, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v216, |v216|, v243, v216\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v244, v215, v[224:227]\n"
      "ds_read_b128 v[252:255], %8 offset:41216\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v245, v215, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v246, v215, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v247, v215, v[236:239]\n"
      "ds_read_b128 v[212:215], %9 offset:832\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v217, |v217|, v243, v217\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v248, v216, v[224:227]\n"
      "ds_read_b128 v[244:247], %8 offset:41472\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v249, v216, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v250, v216, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v251, v216, v[236:239]\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v218, |v218|, v243, v218\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v252, v217, v[224:227]\n"
      "ds_read_b128 v[248:251], %8 offset:41728\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v253, v217, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v254, v217, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v255, v217, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v219, |v219|, v243, v219\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v244, v218, v[224:227]\n"
      "ds_read_b128 v[252:255], %8 offset:45056\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v245, v218, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v246, v218, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v247, v218, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v220, |v220|, v243, v220\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v248, v219, v[224:227]\n"
      "ds_read_b128 v[244:247], %8 offset:45312\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v249, v219, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v250, v219, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v251, v219, v[236:239]\n"
      "ds_read_b128 v[216:219], %9 offset:896\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v221, |v221|, v243, v221\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v252, v220, v[224:227]\n"
      "ds_read_b128 v[248:251], %8 offset:45568\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v253, v220, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v254, v220, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v255, v220, v[236:239]\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v222, |v222|, v243, v222\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v244, v221, v[224:227]\n"
      "ds_read_b128 v[252:255], %8 offset:45824\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v245, v221, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v246, v221, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v247, v221, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v223, |v223|, v243, v223\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v248, v222, v[224:227]\n"
      "ds_read_b128 v[244:247], %8 offset:49152\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v249, v222, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v250, v222, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v251, v222, v[236:239]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[224:227], v252, v223, v[224:227]\n"
      "ds_read_b128 v[248:251], %8 offset:49408\n"
      "v_mfma_f32_16x16x4_f32 v[228:231], v253, v223, v[228:231]\n"
      "v_mfma_f32_16x16x4_f32 v[232:235], v254, v223, v[232:235]\n"
      "v_mfma_f32_16x16x4_f32 v[236:239], v255, v223, v[236:239]\n"
      "ds_read_b128 v[220:223], %9 offset:960\n"
      "s_nop 15\n"
      "v_fma_f32 v224, |v224|, v243, v224\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_fma_f32 v225, |v225|, v243, v225\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v244, v224, v[208:211]\n"
      "ds_read_b128 v[252:255], %8 offset:49664\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v245, v224, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v246, v224, v[216:219]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v247, v224, v[220:223]\n"
      "v_fma_f32 v226, |v226|, v243, v226\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v248, v225, v[208:211]\n"
      "ds_read_b128 v[244:247], %8 offset:49920\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v249, v225, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v250, v225, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v251, v225, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v227, |v227|, v243, v227\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v252, v226, v[208:211]\n"
      "ds_read_b128 v[248:251], %8 offset:53248\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v253, v226, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v254, v226, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v255, v226, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v228, |v228|, v243, v228\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v244, v227, v[208:211]\n"
      "ds_read_b128 v[252:255], %8 offset:53504\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v245, v227, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v246, v227, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v247, v227, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v229, |v229|, v243, v229\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v248, v228, v[208:211]\n"
      "ds_read_b128 v[244:247], %8 offset:53760\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v249, v228, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v250, v228, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v251, v228, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v230, |v230|, v243, v230\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v252, v229, v[208:211]\n"
      "ds_read_b128 v[248:251], %8 offset:54016\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v253, v229, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v254, v229, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v255, v229, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v231, |v231|, v243, v231\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v244, v230, v[208:211]\n"
      "ds_read_b128 v[252:255], %8 offset:57344\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v245, v230, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v246, v230, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v247, v230, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v232, |v232|, v243, v232\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v248, v231, v[208:211]\n"
      "ds_read_b128 v[244:247], %8 offset:57600\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v249, v231, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v250, v231, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v251, v231, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v233, |v233|, v243, v233\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v252, v232, v[208:211]\n"
      "ds_read_b128 v[248:251], %8 offset:57856\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v253, v232, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v254, v232, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v255, v232, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v234, |v234|, v243, v234\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v244, v233, v[208:211]\n"
      "ds_read_b128 v[252:255], %8 offset:58112\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v245, v233, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v246, v233, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v247, v233, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v235, |v235|, v243, v235\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v248, v234, v[208:211]\n"
      "ds_read_b128 v[244:247], %8 offset:61440\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v249, v234, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v250, v234, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v251, v234, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v236, |v236|, v243, v236\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v252, v235, v[208:211]\n"
      "ds_read_b128 v[248:251], %8 offset:61696\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v253, v235, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v254, v235, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v255, v235, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v237, |v237|, v243, v237\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v244, v236, v[208:211]\n"
      "ds_read_b128 v[252:255], %8 offset:61952\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v245, v236, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v246, v236, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v247, v236, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v238, |v238|, v243, v238\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v248, v237, v[208:211]\n"
      "ds_read_b128 v[244:247], %8 offset:62208\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v249, v237, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v250, v237, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v251, v237, v[220:223]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_fma_f32 v239, |v239|, v243, v239\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v252, v238, v[208:211]\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v253, v238, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v254, v238, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v255, v238, v[220:223]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "v_mfma_f32_16x16x4_f32 v[208:211], v244, v239, v[208:211]\n"
      "v_mfma_f32_16x16x4_f32 v[212:215], v245, v239, v[212:215]\n"
      "v_mfma_f32_16x16x4_f32 v[216:219], v246, v239, v[216:219]\n"
      "v_mfma_f32_16x16x4_f32 v[220:223], v247, v239, v[220:223]\n"
      "s_nop 15\n"
      "s_waitcnt lgkmcnt(0)\n"
      : "+{v[208:211]}"(p[0]), "+{v[212:215]}"(p[1]), "+{v[216:219]}"(p[2]), "+{v[220:223]}"(p[3]),
        "=&{v[224:227]}"(q[0]), "=&{v[228:231]}"(q[1]), "=&{v[232:235]}"(q[2]), "=&{v[236:239]}"(q[3])
      : "v"(w_addr), "v"(b_addr)
      : "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
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
#ifndef BGM_NO_ASM_DENSE
    if constexpr (R == 1 && KT == 4 && KSL == 4 && GS == 4) {
      dense_group4_k64_asm(lds_byte_addr(base), in[0], acc[0][T0], acc[0][T0 + 1], acc[0][T0 + 2], acc[0][T0 + 3]);
      dense_groups<T0 + GS, KT, KSL, NT, R>(wl, lane_off, in, acc);
      return;
    }
#endif
    if constexpr (R == 1 && NKS * GS <= BGM_PREFETCH_ALL_MAX) {
      // Small layer (f / h nets, first layers): one or two MFMAs per K-step cannot cover an LDS round trip, and
      // hipcc sinks each fragment load to its first use, so the layer would pay one exposed LDS latency PER STEP
      // (~110 cycles x 16 steps vs 32-64 cycles of MFMA).  Request every fragment of the group up front (the
      // compiler fence keeps the loads above it) and let the counted waits drain them in order: one round trip.
      AFrag<GS> a[NKS];
#pragma unroll
      for (int s = 0; s < NKS; ++s) a[s].load(base + (16 * (s >> 2) + (s & 3)) * 16 * GS);
      BGM_NO_HOIST();
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
#pragma unroll
        for (int u = 0; u < GS; ++u) acc[0][T0 + u] = BGM_MFMA(a[s].get(u), in[0][s >> 2][s & 3], acc[0][T0 + u]);
      }
    } else {
    // software pipeline in SOURCE order: the A fragment of K-step s+1 is requested before the MFMAs of
    // step s (hipcc sinks the load back to its first use; the second wave of the SIMD covers it).
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

// Two independent layers of the SAME shape (the f and h nets of CausalBGM) evaluated in lock step: K-step by K-step
// the MFMAs of both are interleaved, so twice as many independent accumulators are in flight.  Alone, the narrow
// f / h layers are chains of one or two dependent MFMAs per step; with a second wave streaming g-net MFMAs through
// the same matrix pipe those chains measured 3x their solo time (17.1k vs 6.0k cycles per transition).
// Single tile group per layer (NT in {4, 2, 1}); fragments are requested a chunk ahead (<= 32 registers).
template <int KT, int KSL, int NT>
__device__ __forceinline__ void dense_pair(const float *wlA, const float *blA, const float *wlB, const float *blB, int lane_off, int g,
                                           const f32x4 (&inA)[1][KT], const f32x4 (&inB)[1][KT], f32x4 (&accA)[1][NT],
                                           f32x4 (&accB)[1][NT]) {
  static_assert(NT == 4 || NT == 2 || NT == 1, "dense_pair: one tile group per layer");
  constexpr int GS = NT, NKS = 4 * (KT - 1) + KSL;
  constexpr int CH = (NKS * GS <= 16) ? NKS : (16 / GS);
  bias_init<NT, 1>(blA, g, accA);
  bias_init<NT, 1>(blB, g, accB);
  const float *baseA = wlA + lane_off * GS, *baseB = wlB + lane_off * GS;
#pragma unroll
  for (int c0 = 0; c0 < NKS; c0 += CH) {
    AFrag<GS> fa[CH], fb[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int s = c0 + i;
      if (s < NKS) {
        fa[i].load(baseA + (16 * (s >> 2) + (s & 3)) * 16 * GS);
        fb[i].load(baseB + (16 * (s >> 2) + (s & 3)) * 16 * GS);
      }
    }
    BGM_NO_HOIST();
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int s = c0 + i;
      if (s < NKS) {
#pragma unroll
        for (int u = 0; u < GS; ++u) {
          accA[0][u] = BGM_MFMA(fa[i].get(u), inA[0][s >> 2][s & 3], accA[0][u]);
          accB[0][u] = BGM_MFMA(fb[i].get(u), inB[0][s >> 2][s & 3], accB[0][u]);
        }
      }
    }
  }
}

template <int NT, int R>
__device__ __forceinline__ void lrelu_s_inplace(f32x4 (&a)[R][NT]) {
#pragma unroll
  for (int rr = 0; rr < R; ++rr)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) a[rr][t][r] = lrelu_s(a[rr][t][r]);
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

// Same copy with eight 16-byte loads in flight per thread, for kernels whose run time is the copy: copied one float4 at
// a time (load, wait, store) the 157 KB blob takes ~25 us per block, which is most of the B = 32 fit kernels (two row
// tiles = a few microseconds of MFMA work).  Kept separate from lds_fill: hipcc 7.2 fails ("Illegal instruction
// detected: Operand has incorrect register class") when this body is inlined into the persistent sampling kernels.
__device__ __forceinline__ void lds_fill_fast(float *lds, const float *blob, int total_floats) {
  const f32x4 *src = reinterpret_cast<const f32x4 *>(blob);
  f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
  const int n4 = total_floats / 4, stride = blockDim.x;
  int i = threadIdx.x;
  for (; i + 7 * stride < n4; i += 8 * stride) {
    f32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) dst[i + u * stride] = t[u];
  }
  for (; i < n4; i += stride) dst[i] = src[i];
  __syncthreads();
}
// copy a packed weight blob (global) into LDS, all threads of the block
__device__ __forceinline__ void lds_fill(float *lds, const float *blob, int total_floats) {
  const f32x4 *src = reinterpret_cast<const f32x4 *>(blob);
  f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
  for (int i = threadIdx.x; i < total_floats / 4; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
}
