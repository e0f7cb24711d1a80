// bgm_kernels.h -- BGM posterior path on gfx950: masked log-posterior + gradient, HMC, predictive draws.
//
// replaces (src/bayesgm/models/bgm/base.py):
//   get_log_posterior        :665-705  -> bgm_logp_grad<> (device), bgm_logpost_kernel
//   tfp_mcmc_sampler         :709-830  -> bgm_hmc_kernel (+ bgm_hmc_adapt_kernel for SimpleStepSizeAdaptation)
//   predict_on_posteriors    :511-525  -> bgm_predict_cells_kernel
// g_net = BaseVariationalNet (networks/base.py:53-117) in inference mode: the input BatchNorm is an
// affine map and is folded into the first Dense layer on the host.
//
// HMC alternates forward and backward passes, so both orientations of every weight matrix are
// needed in the same kernel; two packed copies exceed the 160 KiB LDS.  A SINGLE copy is kept in a
// dual-access layout  [out tile][in row][17]  (row stride 17 floats): the forward A fragment
// (lanes = 16 consecutive outputs of one input row) and the backward A fragment (lanes = 16
// consecutive input rows of one output column) are both <= 2-way bank conflicted ds_read_b32.
// The first layer is stored slot-permuted (slot 16t+4g+r holds input feature 16t+4r+g) so that the
// state z, the momentum and dlogp/dz all live in the layout the Philox spec fills with one call per
// lane (feature 16t + 4r + g in register r of lane group g).
#pragma once
#include "bgm_device.h"

// One head "tile pair": the mean tile and the variance tile of one 16-feature block, contiguous.
#define BGM_PAIR (2 * 64 * 17)

struct BgmMeta {
  int q, p, n_hh;        // latent dim, data dim, hidden->hidden layers of the trunk
  int ntx;               // 16-feature blocks of the data dimension
  int w1, b1;            // [4 tiles][16*KTQ][17], [64]
  int wh, bh;            // n_hh x ([4][64][17]), n_hh x [64]
  int bhd;               // head biases [2][16*ntx]  (mean then var)
  int whd;               // head weights [ntx][BGM_PAIR]  (LAST: the wide kernels keep them out of LDS)
  int total;             // floats in the blob
  int lds_resident;      // floats copied into LDS at kernel start: total (resident heads) or whd (wide)
  int stage;             // wide: LDS offset of the 2 x BGM_PAIR staging buffers
};

// Wide data rows (x_dim beyond the LDS-resident variants, e.g. BASELINE config C4, p = 500): the 2 x 64 x p
// head weights stay in HBM/L2 and every block streams them through a double-buffered LDS stage, one
// tile pair at a time, shared by all of its waves: the block walks the head tiles in lock step (one
// barrier per tile), so a tile pair is fetched once per block and gradient evaluation instead of
// once per wave.  The data row is re-read per tile (L2) instead of living in 4*ntx registers.
struct BgmHeadStream {
  const float *src;      // global: [ntx][BGM_PAIR]
  float *buf;            // LDS:    [2][BGM_PAIR]
  int cur, tid, nthreads;
  f32x4 r0, r1;
  __device__ __forceinline__ void fetch(int tx) {
    const f32x4 *s = reinterpret_cast<const f32x4 *>(src + (long long)tx * BGM_PAIR);
    if (tid < BGM_PAIR / 4) r0 = s[tid];
    if (tid + nthreads < BGM_PAIR / 4) r1 = s[tid + nthreads];
  }
  __device__ __forceinline__ void commit() {   // publish the fetched pair, flip buffers (block-wide barrier)
    f32x4 *d = reinterpret_cast<f32x4 *>(buf + (cur ^ 1) * BGM_PAIR);
    if (tid < BGM_PAIR / 4) d[tid] = r0;
    if (tid + nthreads < BGM_PAIR / 4) d[tid + nthreads] = r1;
    __syncthreads();
    cur ^= 1;
  }
  __device__ __forceinline__ const float *tile() const { return buf + cur * BGM_PAIR; }
  __device__ __forceinline__ void begin(const float *blob, const BgmMeta &m, float *lds) {
    src = blob + m.whd; buf = lds + m.stage; tid = threadIdx.x; nthreads = blockDim.x;
    cur = 1;
    fetch(0);
    commit();            // tile 0 is current on entry to every gradient evaluation
  }
};

// forward: acc[to] += W^T in   (acc pre-initialised with the bias).  The A values of K-step s+1 are
// loaded while the MFMAs of step s issue; a compiler memory fence per step stops hipcc from
// clustering all 16*KT*NT independent ds_read_b32 ahead of the MFMAs (register blow-up).
template <int KT, int NT>
__device__ __forceinline__ void fwd17(const float *wl, int j, int g, const f32x4 (&in)[KT], f32x4 (&acc)[NT]) {
  BGM_OPAQUE2(j, g);
  constexpr int KR = 16 * KT;
  float a_cur[NT], a_nxt[NT];
  const float *base = wl + (4 * g) * 17 + j;
#pragma unroll
  for (int to = 0; to < NT; ++to) a_cur[to] = base[to * KR * 17];
#pragma unroll
  for (int s = 0; s < 4 * KT; ++s) {
    const int t = s >> 2, r = s & 3;
    if (s + 1 < 4 * KT) {
      const float *row = base + (16 * ((s + 1) >> 2) + ((s + 1) & 3)) * 17;
#pragma unroll
      for (int to = 0; to < NT; ++to) a_nxt[to] = row[to * KR * 17];
    }
#pragma unroll
    for (int to = 0; to < NT; ++to) acc[to] = BGM_MFMA(a_cur[to], in[t][r], acc[to]);
    BGM_NO_HOIST();
#pragma unroll
    for (int to = 0; to < NT; ++to) a_cur[to] = a_nxt[to];
  }
}
// backward to the input: out[ti] += W dpre ;  KT_IN input tiles, NT_K output (contraction) tiles
template <int KT_IN, int NT_K>
__device__ __forceinline__ void bwd17(const float *wl, int i, int g, const f32x4 (&dpre)[NT_K], f32x4 (&out)[KT_IN]) {
  BGM_OPAQUE2(i, g);
  constexpr int KR = 16 * KT_IN;
  float a_cur[KT_IN], a_nxt[KT_IN];
  const float *base = wl + i * 17 + 4 * g;
#pragma unroll
  for (int ti = 0; ti < KT_IN; ++ti) a_cur[ti] = base[16 * ti * 17];
#pragma unroll
  for (int s = 0; s < 4 * NT_K; ++s) {
    const int t = s >> 2, r = s & 3;
    if (s + 1 < 4 * NT_K) {
      const float *col = base + ((s + 1) >> 2) * KR * 17 + ((s + 1) & 3);
#pragma unroll
      for (int ti = 0; ti < KT_IN; ++ti) a_nxt[ti] = col[16 * ti * 17];
    }
#pragma unroll
    for (int ti = 0; ti < KT_IN; ++ti) out[ti] = BGM_MFMA(a_cur[ti], dpre[t][r], out[ti]);
    BGM_NO_HOIST();
#pragma unroll
    for (int ti = 0; ti < KT_IN; ++ti) a_cur[ti] = a_nxt[ti];
  }
}
template <int NT>
__device__ __forceinline__ void bias17(const float *bl, int g, f32x4 (&acc)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = *reinterpret_cast<const f32x4 *>(bl + 16 * t + 4 * g);
}

// mean tile (at wl) and variance tile (at wl + 64*17) of one 16-feature block: forward ...
__device__ __forceinline__ void heads_fwd17(const float *wl, int j, int g, const f32x4 (&h)[4], f32x4 (&ms)[2]) {
  BGM_OPAQUE2(j, g);
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
    ms[0] = BGM_MFMA(c0, h[t][r], ms[0]);
    ms[1] = BGM_MFMA(c1, h[t][r], ms[1]);
    BGM_NO_HOIST();
    c0 = n0; c1 = n1;
  }
}
// ... and backward: dh[ti] += Wmean dmu + Wvar ds
__device__ __forceinline__ void heads_bwd17(const float *wl, int i, int g, const f32x4 (&dms)[2], f32x4 (&dh)[4]) {
  BGM_OPAQUE2(i, g);
  const float *base = wl + i * 17 + 4 * g;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float am[4], av[4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) { am[ti] = base[16 * ti * 17 + r]; av[ti] = base[64 * 17 + 16 * ti * 17 + r]; }
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      dh[ti] = BGM_MFMA(am[ti], dms[0][r], dh[ti]);
      dh[ti] = BGM_MFMA(av[ti], dms[1][r], dh[ti]);
    }
    BGM_NO_HOIST();
  }
}

// The data rows of the 16 chains of a wave: registers for the LDS-resident variants (NTX > 0), a
// pointer to the (clamped) row for the wide variant (NTX == 0; the row is re-read per head tile).
template <int NTX> struct BgmX { f32x4 r[NTX]; };
template <> struct BgmX<0> { const float *row; };

// one 16-feature head block: forward, masked NLL, (optionally) dlogp/d(mean, s_raw) and backward into dh
template <bool WANT_GRAD>
__device__ __forceinline__ void bgm_head_tile(const float *wl, const float *lds, const BgmMeta &m, int tx, int j, int g,
                                              const f32x4 (&h)[4], const f32x4 &xv, float &nll, f32x4 (&dh)[4]) {
  f32x4 ms[2];
  ms[0] = *reinterpret_cast<const f32x4 *>(lds + m.bhd + 16 * tx + 4 * g);
  ms[1] = *reinterpret_cast<const f32x4 *>(lds + m.bhd + 16 * (m.ntx + tx) + 4 * g);
  heads_fwd17(wl, j, g, h, ms);
  f32x4 dms[2];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float x_ = xv[r];
    const bool obs = (x_ == x_) && (16 * tx + 4 * g + r < m.p);   // NaN = missing
    const float s2 = softplus_f(ms[1][r]) + BGM_EPS;
    const float inv = fast_rcp(s2);
    const float d = obs ? x_ - ms[0][r] : 0.0f;
    nll += obs ? 0.5f * (d * d * inv + fast_log(s2)) : 0.0f;
    if (WANT_GRAD) {
      const float sg = fast_rcp(1.0f + fast_exp(-ms[1][r]));      // sigmoid(s) = d softplus / ds
      dms[0][r] = d * inv;                                                        // dlogp/dmu
      dms[1][r] = obs ? (0.5f * d * d * inv * inv - 0.5f * inv) * sg : 0.0f;     // dlogp/ds
    }
  }
  if (WANT_GRAD) heads_bwd17(wl, j, g, dms, dh);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Split precision (opt-in, bgm_bgm_set_precision(h, 2) / params['hmc_precision'] = 'f16x3'; PREC = 2 below).  Every product of the
// generator -- the two p-wide heads (~80 % of a gradient evaluation's matrix work at BASELINE config C4: 64 000 of 81 024 MACs) and the
// trunk -- runs on v_mfma_f32_16x16x32_f16 (K = 32 per instruction, half the cycles of the K = 4 fp32 one) with every operand a sum of
// two fp16 numbers, x = x_hi + x_lo, and  W h ~= W_lo h_hi + W_hi h_lo + W_hi h_hi  (fp32 accumulation; the dropped term is 2^-22 of
// the product): a 16-feature head block is 24 matrix instructions instead of 64, a 64 -> 64 layer 24 instead of 64.
//   forward   out(16 features) = W^T h:  M = output feature, K = 64 units in two blocks of 32; lane group g supplies k-slot
//             u = 4 s + r  <->  unit 16 (2 b + s) + 4 g + r, i.e. a layer's accumulators ARE the next product's B operand (split once per layer);
//   backward  d_in(64 units) += W d_out: M = input unit, K = the outputs in blocks of 32 (heads: the block's 16 mean + 16 variance
//             outputs, lane group g supplies slot u: u < 4 -> mean output 4 g + u, else variance output 4 g + u - 4): again the lane's own registers.
// NO weight lives in LDS.  Forward and backward need both orientations of every matrix; as packed fp16 fragments that is 2 x 16 KiB per
// hidden layer, which does not fit beside a stage.  So the whole generator is ONE LINEAR STREAM of 32 KiB steps, packed on the host
// (bgm_api.hip) and walked by the workgroup in lock step through a double-buffered LDS stage filled by global_load_lds_dwordx4 (the
// fragments are lane-linear: exactly what that instruction writes; no staging registers, no ds_write pass):
//     [L1 (forward + transposed), hidden 1 .. NH - 1 forward | head blocks | hidden NH - 1 .. 1 transposed, L1]   two 16 KiB units per step
// 2 ceil(NH / 2) + ceil(ntx / 2) steps per gradient evaluation (22 at NH = 5, p = 500: 0.7 MB, L2 resident); LDS holds the biases and the 64 KiB stage.
// The likelihood arithmetic, the leapfrog and all accumulations stay fp32.
// fp16 range: a weight beyond 65504 is clamped by the packer; dlogp/d(mean, s) and the back-propagated d(pre-activation) beyond 6e4 (a
// variance ~1e-5 under a unit residual) are clamped in the kernel -- the fp32 kernels have no such bound.
// Measured (N = 2e5, p = 500, 10 leapfrog steps, one MI355X): 7.3 -> 3.2 ms per transition (2.26x); heads only, trunk on the fp32
// dual-access layout: 4.1 ms.  What bounds it: per wave the heads' likelihood arithmetic (~35 VALU instructions per (row, feature), four
// of them transcendental) costs as many cycles as their 24 matrix instructions per block (profiles/r06_bgm_hmc_f16x3_*.txt).
// ---------------------------------------------------------------------------------------------------------------------------
typedef _Float16 bgm_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 bgm_h2 __attribute__((ext_vector_type(2)));
typedef float bgm_f2 __attribute__((ext_vector_type(2)));
typedef unsigned bgm_u4 __attribute__((ext_vector_type(4)));
#define BGM_X3_BLOCK_BYTES 16384
#define BGM_MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void bgm_split_pair(float a, float b, unsigned &hi, unsigned &lo) {
  const bgm_h2 h = __builtin_convertvector(bgm_f2{a, b}, bgm_h2);
  hi = __builtin_bit_cast(unsigned, h);
  const bgm_h2 l = __builtin_convertvector(bgm_f2{a - (float)h[0], b - (float)h[1]}, bgm_h2);      // (exact subtractions)
  lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void bgm_split8(const f32x4 &a, const f32x4 &b, bgm_h8 &hi, bgm_h8 &lo) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  bgm_split_pair(a[0], a[1], h0, l0);
  bgm_split_pair(a[2], a[3], h1, l1);
  bgm_split_pair(b[0], b[1], h2, l2);
  bgm_split_pair(b[2], b[3], h3, l3);
  hi = __builtin_bit_cast(bgm_h8, bgm_u4{h0, h1, h2, h3});
  lo = __builtin_bit_cast(bgm_h8, bgm_u4{l0, l1, l2, l3});
}

// BGM_X3_STEP head blocks per step of the stream (one barrier per step): with the matrix work of a block down to ~400 cycles a
// barrier and a fetch per 16 features were what the waves waited for (34 % of the wave cycles parked, r06_pmc_sq_bgm_hmc_f16x3.txt)
#ifndef BGM_X3_STEP
#define BGM_X3_STEP 2
#endif
#ifndef BGM_X3_INTERLEAVE
#define BGM_X3_INTERLEAVE 0
#endif
#ifndef BGM_X3_DMA
#define BGM_X3_DMA 1      // 1: the stage is filled by global_load_lds_dwordx4 (no staging registers, no ds_write pass); 0: through registers
#endif
// X4 (compile time): x_dim % 4 == 0, the data rows are 16-byte aligned -- one request per lane and block; else four clamped 4-byte
// requests.  A run-time branch between the two forms inside the loop made hipcc guard the destination registers of one form against
// the other's requests with counted waits that, at run time, fall on the direct-to-LDS loads in flight.
template <int WAVES, bool X4 = false>
struct BgmHeadStreamX3 {
  static constexpr int STEP_VEC = BGM_X3_STEP * BGM_X3_BLOCK_BYTES / 16, NT = 64 * WAVES, K = (STEP_VEC + NT - 1) / NT;
  const f32x4 *src;      // global: [steps][BGM_X3_STEP][BGM_X3_BLOCK_BYTES]
  unsigned char *buf;    // LDS:    [2][BGM_X3_STEP][BGM_X3_BLOCK_BYTES]
  int cur, tid;
#if BGM_X3_DMA
  // every lane's 16 bytes go straight from L2 to the stage: destination = wave-uniform base + lane x 16 (the fragments are lane-linear,
  // which is exactly the layout the instruction writes); the other buffer is the one nobody reads during this step.
  // Issued as inline assembly: behind __builtin_amdgcn_global_load_lds hipcc (7.2) places s_waitcnt vmcnt(0) in front of the NEXT LDS
  // read of any address (it cannot tell the two buffers of the stage apart), i.e. every step began by waiting out the latency of
  // the loads it had just issued -- and of the data-row request beside them.  The wait that is needed is the one in commit().
  __device__ __forceinline__ void fetch(int step) {
#if defined(BGM_X3_ABL_NOSTREAM) || defined(BGM_X3_ABL_NODMA)      // (NODMA: the barriers stay, the stage is never refilled; timing only)
    return;
#endif
    const f32x4 *s = src + (long long)step * STEP_VEC + tid;
    typedef __attribute__((address_space(3))) unsigned char *lds_ptr;
    const unsigned dst0 = (unsigned)(unsigned long long)(lds_ptr)(buf + (cur ^ 1) * (BGM_X3_STEP * BGM_X3_BLOCK_BYTES)) + (unsigned)(tid & ~63) * 16u;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (K * NT == STEP_VEC || (tid & ~63) + k * NT < STEP_VEC) {
        const unsigned m0v = __builtin_amdgcn_readfirstlane(dst0 + (unsigned)(k * NT) * 16u);
        unsigned m0_saved;      // (M0 is handed back as it was found: the compiler does not model a clobber of it)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_saved) : "s"(m0v), "v"(s + k * NT) : "memory");
      }
  }
  __device__ __forceinline__ void commit() {
#ifdef BGM_X3_ABL_NOSTREAM
    return;
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the next step have landed ...
    __syncthreads();                                         // ... and everybody's
    cur ^= 1;
  }
#else
  f32x4 r[K];
  __device__ __forceinline__ void fetch(int step) {      // step = first block / BGM_X3_STEP
#ifdef BGM_X3_ABL_NOSTREAM
    return;
#endif
    const f32x4 *s = src + (long long)step * STEP_VEC;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (K * NT == STEP_VEC || tid + k * NT < STEP_VEC) r[k] = s[tid + k * NT];
  }
  __device__ __forceinline__ void commit() {
#ifdef BGM_X3_ABL_NOSTREAM      // development ablation: no refill of the stage, no barrier (wrong results: timing only)
    return;
#endif
    f32x4 *d = reinterpret_cast<f32x4 *>(buf + (cur ^ 1) * (BGM_X3_STEP * BGM_X3_BLOCK_BYTES));
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (K * NT == STEP_VEC || tid + k * NT < STEP_VEC) d[tid + k * NT] = r[k];
    __syncthreads();
    cur ^= 1;
  }
#endif
  __device__ __forceinline__ const unsigned char *tile(int b) const { return buf + (cur * BGM_X3_STEP + b) * BGM_X3_BLOCK_BYTES; }
  // The data values of a step are requested ONE STEP AHEAD (xn; the first step of an evaluation by the last step of the previous
  // one: the row does not change within a launch's tile): a block's products are ~200 cycles now, the rows come from HBM (1.25 GB at
  // C4's share, re-read per evaluation) -- requested at the block's start they were what every block waited for.
  f32x4 xn[BGM_X3_STEP];
  bool x_valid;
  // (unconditional requests at clamped addresses: a column at or beyond p is masked where the value is used -- bgm_x3_lik's in_range --
  // so no zero fill, no branch around a request: hipcc counts the requests and places no wait between them)
  __device__ __forceinline__ void load_x(const float *row, int p, int step, int g, f32x4 (&dst)[BGM_X3_STEP]) const {
#pragma unroll
    for (int b = 0; b < BGM_X3_STEP; ++b) {
      const int c = 16 * (BGM_X3_STEP * step + b) + 4 * g;
      if (X4) {                         // (rows are 16-byte aligned: one request per block and lane)
        dst[b] = *reinterpret_cast<const f32x4 *>(row + (c < p ? c : p - 4));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[b][r] = row[c + r < p ? c + r : p - 1];
      }
    }
  }
  // last step's request as values: the wait for it is placed HERE, in front of this step's requests (the requests sit in branches, so
  // hipcc cannot count them and would otherwise wait for ALL outstanding loads -- this step's included -- at the first use of a value)
  __device__ __forceinline__ f32x4 take_x(int b) const {
    float x0 = xn[b][0], x1 = xn[b][1], x2 = xn[b][2], x3 = xn[b][3];
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    return f32x4{x0, x1, x2, x3};
  }
  __device__ __forceinline__ void load_x1(const float *row, int p, int tx, int g) {      // one block's values into xn[0] (bgmfx_kernels.h: one block per step)
    const int c = 16 * tx + 4 * g;
    if (X4) {
      xn[0] = *reinterpret_cast<const f32x4 *>(row + (c < p ? c : p - 4));
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) xn[0][r] = row[c + r < p ? c + r : p - 1];
    }
  }
  __device__ __forceinline__ void begin_at(const unsigned char *frags, float *stage) {
    src = reinterpret_cast<const f32x4 *>(frags); buf = reinterpret_cast<unsigned char *>(stage); tid = threadIdx.x;
    cur = 1; x_valid = false;
    fetch(0);
    commit();
  }
  __device__ __forceinline__ void begin(const unsigned char *frags, const BgmMeta &m, float *lds) { begin_at(frags, lds + m.stage); }
};

// One 16-feature head block in split precision, in three parts so that a step of the stream (BGM_X3_STEP blocks) can be issued as
// fwd(0), fwd(1), epilogue(0), bwd(0), epilogue(1), bwd(1): the likelihood arithmetic of one block (~140 VALU instructions, six of
// them transcendental) then sits under the matrix instructions of the other instead of between its own two products.  All waves of
// a workgroup are in phase (one barrier per step), so without this every wave of a SIMD wants the matrix pipe, then the VALU, at the same time.
// hh / hl: the trunk output split into its two K blocks (once per evaluation).
__device__ __forceinline__ void bgm_x3_fwd(const unsigned char *tile, const float *lds, const BgmMeta &m, int tx, int lane, int g,
                                           const bgm_h8 (&hh)[2], const bgm_h8 (&hl)[2], f32x4 (&part)[4]) {
  const bgm_h8 *fr = reinterpret_cast<const bgm_h8 *>(tile) + lane;      // fragment f at fr[64 f]
  const int tb = tx < m.ntx ? tx : m.ntx - 1;                             // (a padding block behind an odd block count reads the last bias)
  // (head, K block) = four INDEPENDENT accumulator chains of three products each, issued round-robin: a product never follows
  // its own accumulator's previous one (the K blocks are summed in the epilogue: 8 additions)
  bgm_h8 ah[4], al[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) { ah[f] = fr[64 * (2 * f)]; al[f] = fr[64 * (2 * f + 1)]; }
  part[0] = *reinterpret_cast<const f32x4 *>(lds + m.bhd + 16 * tb + 4 * g);
  part[2] = *reinterpret_cast<const f32x4 *>(lds + m.bhd + 16 * (m.ntx + tb) + 4 * g);
  part[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; part[3] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int f = 0; f < 4; ++f) part[f] = BGM_MFMA_H(al[f], hh[f & 1], part[f]);
#pragma unroll
  for (int f = 0; f < 4; ++f) part[f] = BGM_MFMA_H(ah[f], hl[f & 1], part[f]);
#pragma unroll
  for (int f = 0; f < 4; ++f) part[f] = BGM_MFMA_H(ah[f], hh[f & 1], part[f]);
}
// the Gaussian likelihood of one (row, feature) from the heads' raw outputs (mean, s): variance = softplus(s) + eps; dmu / ds clamped
// into the fp16 range (they are the next products' operands)
template <bool WANT_GRAD>
__device__ __forceinline__ void bgm_x3_lik(float x_, float mu_, float s_, bool in_range, bool want_lp, float &nll, float &dmu, float &ds) {
#ifdef BGM_X3_ABL_NOEPI         // development ablation: the likelihood arithmetic replaced by two subtractions (timing only)
  dmu = x_ - mu_; ds = s_ - x_; nll += mu_;
  return;
#endif
  const bool obs = (x_ == x_) && in_range;   // NaN = missing; padding columns / blocks are never observed
  // softplus and its derivative from ONE exponential: e = exp(-|s|), softplus = max(s, 0) + log1p(e), sigmoid = (s >= 0 ? 1 : e) / (1 + e)
  const float e = fast_exp(-fabsf(s_));
  const float l1p = (e < 2.44140625e-4f) ? e * (1.0f - 0.5f * e) : fast_log(1.0f + e);
  const float s2 = vmax(s_, 0.0f) + l1p + BGM_EPS;
  const float inv = fast_rcp(s2);
  const float d = obs ? x_ - mu_ : 0.0f;
  if (want_lp) nll += obs ? 0.5f * (d * d * inv + fast_log(s2)) : 0.0f;
  if (WANT_GRAD) {
    const float sg = (s_ >= 0.0f ? 1.0f : e) * fast_rcp(1.0f + e);
    const float di = d * inv;
    dmu = __builtin_amdgcn_fmed3f(di, -6.0e4f, 6.0e4f);                                              // dlogp/dmu
    ds = obs ? __builtin_amdgcn_fmed3f((0.5f * di * di - 0.5f * inv) * sg, -6.0e4f, 6.0e4f) : 0.0f;   // dlogp/ds
  }
}
template <bool WANT_GRAD>
__device__ __forceinline__ void bgm_x3_epilogue(const BgmMeta &m, int tx, int g, const f32x4 (&part)[4], const f32x4 &xv, bool want_lp,
                                                float &nll, bgm_h8 &dhi, bgm_h8 &dlo) {
  f32x4 dms[2];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float dmu, ds;
    bgm_x3_lik<WANT_GRAD>(xv[r], part[0][r] + part[1][r], part[2][r] + part[3][r], 16 * tx + 4 * g + r < m.p, want_lp, nll, dmu, ds);
    dms[0][r] = dmu; dms[1][r] = ds;
  }
  if (WANT_GRAD) bgm_split8(dms[0], dms[1], dhi, dlo);
}
__device__ __forceinline__ void bgm_x3_bwd(const unsigned char *tile, int lane, const bgm_h8 &dhi, const bgm_h8 &dlo, f32x4 (&dh)[4]) {
  const bgm_h8 *fr = reinterpret_cast<const bgm_h8 *>(tile) + lane;
  bgm_h8 ah[4], al[4];      // hidden tile ti
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) { ah[ti] = fr[64 * (8 + 2 * ti)]; al[ti] = fr[64 * (9 + 2 * ti)]; }
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) dh[ti] = BGM_MFMA_H(al[ti], dhi, dh[ti]);
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) dh[ti] = BGM_MFMA_H(ah[ti], dlo, dh[ti]);
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) dh[ti] = BGM_MFMA_H(ah[ti], dhi, dh[ti]);
}

// log p(z | x_obs) and dlogp/dz for 16 chains held by one wave.
//   z  : feature 16t + 4r + g in register r of tile t
//   xs : data row, feature 16t + 4g + r, NaN = missing (ignored), zero padded beyond p
//   hs : head-weight stream of the block (wide variant only; every wave of the block must call this
//        function the same number of times)
// logp is replicated over the lane groups; grad has the layout of z.
// PREC 1: the heads in split precision (wide variant only, HS = BgmHeadStreamX3); want_lp = false (PREC 1 only): the value of the log
// posterior is not needed (the inner leapfrog steps of an HMC transition use the gradient alone), logp is then undefined.
template <int KTQ, int NTX, int NH, bool WANT_GRAD, int PREC = 0, class HS = BgmHeadStream>
__device__ __forceinline__ void bgm_logp_grad(const float *lds, const BgmMeta &m, int j, int g,
                                              const f32x4 (&z)[KTQ], const BgmX<NTX> &xs, HS &hs,
                                              float &logp, f32x4 (&grad)[KTQ], bool want_lp = true) {
  static_assert(PREC == 0 || NTX == 0, "split precision: the streamed (wide) variant");
  // trunk forward; only the SIGN of every activation is kept for the backward pass (bit 4t+r of sgn[l])
  unsigned sgn[NH];
  f32x4 h[4];
  if constexpr (PREC == 2) {
    // PREC 2: the trunk in split precision too, its weights STREAMED like the head blocks (no trunk weight lives in LDS: one unit of
    // 16 KiB per layer and direction -- forward fragments on the way up, transposed fragments on the way down -- so the workgroup
    // needs 70 KiB instead of 144 and two of them share a CU, out of phase).  Steps of an evaluation: L1, hidden 1 .. NH - 1, the
    // head steps, hidden NH - 1 .. 1 backward, L1 backward; step 0 is current on entry, the last step fetches it for the next call.
    static_assert(KTQ == 1, "split-precision trunk: z_dim <= 16");
    const int lane2 = 16 * g + j;
    constexpr int U = BGM_X3_STEP;      // layer l of the trunk is unit l % U of step l / U: U layers per step, one barrier per U layers
    hs.fetch(1);
    {
      bgm_h8 zh, zl;
      bgm_split8(z[0], f32x4{0.0f, 0.0f, 0.0f, 0.0f}, zh, zl);      // k-slot u < 4 <-> latent feature 4 u + g, the upper half of the K block is zero
      const bgm_h8 *fr = reinterpret_cast<const bgm_h8 *>(hs.tile(0)) + lane2;
      bgm_h8 ah[4], al[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) { ah[mt] = fr[64 * (2 * mt)]; al[mt] = fr[64 * (2 * mt + 1)]; h[mt] = *reinterpret_cast<const f32x4 *>(lds + m.b1 + 16 * mt + 4 * g); }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) h[mt] = BGM_MFMA_H(al[mt], zh, h[mt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) h[mt] = BGM_MFMA_H(ah[mt], zl, h[mt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) h[mt] = BGM_MFMA_H(ah[mt], zh, h[mt]);
    }
    sgn[0] = 0u;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sgn[0] |= (h[t][r] > 0.0f) ? (1u << (4 * t + r)) : 0u;
        h[t][r] = lrelu(h[t][r]);
      }
    asm volatile("" : "+v"(sgn[0]));
    if (U == 1 || NH == 1) hs.commit();
#pragma unroll
    for (int l = 1; l < NH; ++l) {
      BGM_NO_HOIST();
      if (l % U == 0) hs.fetch(l / U + 1);
      bgm_h8 bh_[2], bl_[2];
      bgm_split8(h[0], h[1], bh_[0], bl_[0]);
      bgm_split8(h[2], h[3], bh_[1], bl_[1]);
      const bgm_h8 *fr = reinterpret_cast<const bgm_h8 *>(hs.tile(l % U)) + lane2;
      f32x4 h2[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) h2[mt] = *reinterpret_cast<const f32x4 *>(lds + m.bh + (l - 1) * 64 + 16 * mt + 4 * g);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        bgm_h8 ah[4], al[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { ah[mt] = fr[64 * (2 * (2 * mt + b))]; al[mt] = fr[64 * (2 * (2 * mt + b) + 1)]; }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) h2[mt] = BGM_MFMA_H(al[mt], bh_[b], h2[mt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) h2[mt] = BGM_MFMA_H(ah[mt], bl_[b], h2[mt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) h2[mt] = BGM_MFMA_H(ah[mt], bh_[b], h2[mt]);
      }
      sgn[l] = 0u;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sgn[l] |= (h2[t][r] > 0.0f) ? (1u << (4 * t + r)) : 0u;
          h[t][r] = lrelu(h2[t][r]);
        }
      asm volatile("" : "+v"(sgn[l]));
      if (l % U == U - 1 || l == NH - 1) hs.commit();
    }
  } else {
  bias17<4>(lds + m.b1, g, h);
  fwd17<KTQ, 4>(lds + m.w1, j, g, z, h);
  sgn[0] = 0u;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sgn[0] |= (h[t][r] > 0.0f) ? (1u << (4 * t + r)) : 0u;
      h[t][r] = lrelu(h[t][r]);
    }
  asm volatile("" : "+v"(sgn[0]));   // materialise the mask now (else the activations stay live until backward)
#pragma unroll
  for (int l = 1; l < NH; ++l) {
    BGM_NO_HOIST();
    f32x4 h2[4];
    bias17<4>(lds + m.bh + (l - 1) * 64, g, h2);
    fwd17<4, 4>(lds + m.wh + (l - 1) * (4 * 64 * 17), j, g, h, h2);
    sgn[l] = 0u;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sgn[l] |= (h2[t][r] > 0.0f) ? (1u << (4 * t + r)) : 0u;
        h[t][r] = lrelu(h2[t][r]);
      }
    asm volatile("" : "+v"(sgn[l]));
  }
  }
  // heads, one 16-feature block at a time
  float nll = 0.0f;
  f32x4 dh[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) dh[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr (NTX > 0) {
#pragma unroll
    for (int tx = 0; tx < NTX; ++tx) {
      BGM_NO_HOIST();
      bgm_head_tile<WANT_GRAD>(lds + m.whd + tx * BGM_PAIR, lds, m, tx, j, g, h, xs.r[tx], nll, dh);
    }
  } else if constexpr (PREC >= 1) {
    bgm_h8 hh[2], hl[2];
    bgm_split8(h[0], h[1], hh[0], hl[0]);
    bgm_split8(h[2], h[3], hh[1], hl[1]);
    const int lane = 16 * g + j;
    const int n_steps = (m.ntx + BGM_X3_STEP - 1) / BGM_X3_STEP;
    // (PREC 2: the head steps sit behind the NH forward steps of the trunk; behind the last one comes the first backward step, or
    // step 0 again when no gradient is wanted)
    constexpr int FT = (NH + BGM_X3_STEP - 1) / BGM_X3_STEP;      // steps of the trunk per direction
    const int s0 = PREC == 2 ? FT : 0, s_after = (PREC == 2 && WANT_GRAD) ? FT + n_steps : 0;
#pragma unroll 1
    for (int st = 0; st < n_steps; ++st) {
      BGM_NO_HOIST();
      const int next_step = st + 1 < n_steps ? s0 + st + 1 : s_after;
      // the step's blocks as one straight-line region: forward products of every block first, then per block its likelihood
      // arithmetic (under the products still in flight) and its backward products (an odd block count ends on a padding block of
      // zero fragments whose columns are never observed: it contributes exact zeros)
      f32x4 xv[BGM_X3_STEP], part[BGM_X3_STEP][4];
      if (!hs.x_valid) { hs.load_x(xs.row, m.p, st, g, hs.xn); hs.x_valid = true; }      // (the first evaluation of a tile only)
#pragma unroll
      for (int b = 0; b < BGM_X3_STEP; ++b) xv[b] = hs.take_x(b);
      hs.fetch(next_step);      // (behind take_x: last step's data values are in registers before this step's loads are issued)
      hs.load_x(xs.row, m.p, st + 1 < n_steps ? st + 1 : 0, g, hs.xn);
      asm volatile("" ::: "memory");      // (hipcc sinks a load to just above its first use: the request stays here, a step ahead)
#if BGM_X3_INTERLEAVE
#pragma unroll
      for (int b = 0; b < BGM_X3_STEP; ++b) bgm_x3_fwd(hs.tile(b), lds, m, BGM_X3_STEP * st + b, lane, g, hh, hl, part[b]);
#pragma unroll
      for (int b = 0; b < BGM_X3_STEP; ++b) {
        bgm_h8 dhi, dlo;
        bgm_x3_epilogue<WANT_GRAD>(m, BGM_X3_STEP * st + b, g, part[b], xv[b], want_lp, nll, dhi, dlo);
        if (WANT_GRAD) bgm_x3_bwd(hs.tile(b), lane, dhi, dlo, dh);
      }
#else
#pragma unroll
      for (int b = 0; b < BGM_X3_STEP; ++b) {      // (block after block: 16 accumulator registers fewer in flight; measured equal to the interleaved order)
        bgm_h8 dhi, dlo;
        bgm_x3_fwd(hs.tile(b), lds, m, BGM_X3_STEP * st + b, lane, g, hh, hl, part[0]);
        bgm_x3_epilogue<WANT_GRAD>(m, BGM_X3_STEP * st + b, g, part[0], xv[b], want_lp, nll, dhi, dlo);
        if (WANT_GRAD) bgm_x3_bwd(hs.tile(b), lane, dhi, dlo, dh);
      }
#endif
      hs.commit();
    }
  } else {
#pragma unroll 1
    for (int tx = 0; tx < m.ntx; ++tx) {
      BGM_NO_HOIST();
      hs.fetch(tx + 1 < m.ntx ? tx + 1 : 0);
      f32x4 xv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * tx + 4 * g + r;
        xv[r] = (c < m.p) ? xs.row[c] : 0.0f;
      }
      bgm_head_tile<WANT_GRAD>(hs.tile(), lds, m, tx, j, g, h, xv, nll, dh);
      hs.commit();
    }
  }
  float zsq = 0.0f;
#pragma unroll
  for (int t = 0; t < KTQ; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) zsq = fmaf(z[t][r], z[t][r], zsq);   // padded features are zero
  logp = -sum_over_g(nll + 0.5f * zsq);
  if (WANT_GRAD) {
    if constexpr (PREC == 2) {
      const int lane2 = 16 * g + j;
      constexpr int U = BGM_X3_STEP, FT = (NH + U - 1) / U;
      const int sb0 = FT + (m.ntx + U - 1) / U;      // first backward step; backward unit i = NH - 1 - l is unit i % U of step sb0 + i / U
#pragma unroll
      for (int l = NH - 1; l >= 0; --l) {
        BGM_NO_HOIST();
        const int bi = NH - 1 - l;
        if (bi % U == 0) hs.fetch(bi / U + 1 < FT ? sb0 + bi / U + 1 : 0);
        // d(pre-activation) = dh o LeakyReLU', clamped into the fp16 range like the heads' gradients, split once per layer
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dh[t][r] = __builtin_amdgcn_fmed3f(dh[t][r] * (((sgn[l] >> (4 * t + r)) & 1u) ? 1.0f : BGM_LEAK), -6.0e4f, 6.0e4f);
        bgm_h8 dhh[2], dhl[2];
        bgm_split8(dh[0], dh[1], dhh[0], dhl[0]);
        bgm_split8(dh[2], dh[3], dhh[1], dhl[1]);
        const bgm_h8 *fr = reinterpret_cast<const bgm_h8 *>(hs.tile(bi % U)) + lane2;
        if (l > 0) {
          f32x4 dn[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) dn[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            bgm_h8 ah[4], al[4];
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) { ah[ti] = fr[64 * (2 * (2 * ti + b))]; al[ti] = fr[64 * (2 * (2 * ti + b) + 1)]; }
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) dn[ti] = BGM_MFMA_H(al[ti], dhh[b], dn[ti]);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) dn[ti] = BGM_MFMA_H(ah[ti], dhl[b], dn[ti]);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) dn[ti] = BGM_MFMA_H(ah[ti], dhh[b], dn[ti]);
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) dh[t] = dn[t];
        } else {      // first layer: the latent gradient, rows of the fragment in the latent layout (feature 4 r + g in register r)
          f32x4 ga[2];
          bgm_h8 ah[2], al[2];
#pragma unroll
          for (int b = 0; b < 2; ++b) { ah[b] = fr[64 * (8 + 2 * b)]; al[b] = fr[64 * (9 + 2 * b)]; ga[b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
          for (int b = 0; b < 2; ++b) ga[b] = BGM_MFMA_H(al[b], dhh[b], ga[b]);
#pragma unroll
          for (int b = 0; b < 2; ++b) ga[b] = BGM_MFMA_H(ah[b], dhl[b], ga[b]);
#pragma unroll
          for (int b = 0; b < 2; ++b) ga[b] = BGM_MFMA_H(ah[b], dhh[b], ga[b]);
#pragma unroll
          for (int r = 0; r < 4; ++r) grad[0][r] = ga[0][r] + ga[1][r];
        }
        if (bi % U == U - 1 || l == 0) hs.commit();
      }
    } else {
#pragma unroll
    for (int l = NH - 1; l >= 0; --l) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[t][r] *= ((sgn[l] >> (4 * t + r)) & 1u) ? 1.0f : BGM_LEAK;
      if (l > 0) {
        BGM_NO_HOIST();
        f32x4 dn[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dn[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        bwd17<4, 4>(lds + m.wh + (l - 1) * (4 * 64 * 17), j, g, dh, dn);
#pragma unroll
        for (int t = 0; t < 4; ++t) dh[t] = dn[t];
      }
    }
#pragma unroll
    for (int t = 0; t < KTQ; ++t) grad[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    bwd17<KTQ, 4>(lds + m.w1, j, g, dh, grad);
    }
#pragma unroll
    for (int t = 0; t < KTQ; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) grad[t][r] -= z[t][r];   // prior  -|z|^2/2
  }
}

template <int NTX>
__device__ __forceinline__ void bgm_load_x(const float *x, long long n, int p, long long row, int g, BgmX<NTX> &xs) {
  const float *p_ = x + row * (long long)p;
  if constexpr (NTX > 0) {
#pragma unroll
    for (int t = 0; t < NTX; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * t + 4 * g + r;
        xs.r[t][r] = (c < p) ? p_[c] : 0.0f;
      }
  } else {
    xs.row = p_;
  }
}
// Row tile of this wave in pass `it` of the block.  The wide variant (NTX == 0) needs every wave of a
// block to make the same number of passes (block-wide barriers in the head loop): waves beyond the
// last tile recompute the last row and store nothing.
__device__ __forceinline__ long long bgm_block_passes(long long n_tiles, int waves) {
  const long long per_pass = (long long)gridDim.x * waves;
  return (n_tiles + per_pass - 1) / per_pass;
}
template <int KTQ>
__device__ __forceinline__ void bgm_load_z(const float *z, int q, long long row, int g, f32x4 (&zr)[KTQ]) {
#pragma unroll
  for (int t = 0; t < KTQ; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * t + 4 * r + g;
      zr[t][r] = (f < q) ? z[row * (long long)q + f] : 0.0f;
    }
}
template <int KTQ>
__device__ __forceinline__ void bgm_store_z(float *z, int q, long long row, int g, const f32x4 (&zr)[KTQ]) {
#pragma unroll
  for (int t = 0; t < KTQ; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = 16 * t + 4 * r + g;
      if (f < q) z[row * (long long)q + f] = zr[t][r];
    }
}

template <int PREC, int WAVES, bool X4> struct bgm_stream_of { typedef BgmHeadStream type; };
template <int WAVES, bool X4> struct bgm_stream_of<1, WAVES, X4> { typedef BgmHeadStreamX3<WAVES, X4> type; };
template <int WAVES, bool X4> struct bgm_stream_of<2, WAVES, X4> { typedef BgmHeadStreamX3<WAVES, X4> type; };

// get_log_posterior (+ optional gradient) for n rows
template <int KTQ, int NTX, int NH, int WAVES, int PREC = 0, bool X4 = false>
__global__ __launch_bounds__(64 * WAVES) void bgm_logpost_kernel(const float *blob, BgmMeta m, const float *z,
                                                                 const float *x, long long n, float *out,
                                                                 float *grad_out, const unsigned char *hx3 = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  lds_fill(lds, blob, m.lds_resident);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  using HS = typename bgm_stream_of<PREC, WAVES, X4>::type;
  HS hs;
  if constexpr (PREC >= 1) hs.begin(hx3, m, lds);
  else if constexpr (NTX == 0) hs.begin(blob, m, lds);
  const long long n_tiles = (n + 15) / 16, passes = bgm_block_passes(n_tiles, WAVES);
  for (long long ps = 0; ps < passes; ++ps) {
    BGM_NO_HOIST();
    long long tile = (ps * gridDim.x + blockIdx.x) * WAVES + wave;
    const bool tile_ok = tile < n_tiles;
    if (NTX > 0 && !tile_ok) break;
    tile = tile_ok ? tile : n_tiles - 1;
    long long row = tile * 16 + j;
    const bool ok = tile_ok && row < n;
    row = row < n ? row : n - 1;
    BgmX<NTX> xr;
    f32x4 zr[KTQ], gr[KTQ];
    bgm_load_x<NTX>(x, n, m.p, row, g, xr);
    if constexpr (PREC >= 1) hs.x_valid = false;       // (a new row: nothing of it has been requested ahead)
    bgm_load_z<KTQ>(z, m.q, row, g, zr);
    float lp;
    if (grad_out != nullptr) bgm_logp_grad<KTQ, NTX, NH, true, PREC, HS>(lds, m, j, g, zr, xr, hs, lp, gr);
    else bgm_logp_grad<KTQ, NTX, NH, false, PREC, HS>(lds, m, j, g, zr, xr, hs, lp, gr);
    if (ok) {
      if (g == 0) out[row] = lp;
      if (grad_out != nullptr) bgm_store_z<KTQ>(grad_out, m.q, row, g, gr);
    }
  }
}

// ---------------------------------------------------------------------------
// Hamiltonian Monte Carlo, identity mass, n_leapfrog steps, one chain per row.
// ---------------------------------------------------------------------------
struct BgmHmcKArgs {
  const float *blob;
  const float *x;            // [n x p], NaN = missing
  long long n, row_base;
  float *state, *logp, *grad;   // chain state [n x q], cached logp [n] and dlogp/dz [n x q] (in/out)
  int init, it_begin, n_iters, burn_in, n_leapfrog;
  const float *step;         // device scalar: current step size
  unsigned k0, k1;
  double *acc_prob_sum;      // [it]  += sum over chains of exp(min(0, log_accept_ratio))
  unsigned *acc_count;       // [it]  += accepted chains
  float *draws;              // [n_keep x n x q] or NULL
  BgmMeta m;
  const unsigned char *hx3;  // PREC 1: the packed fp16 head fragments [ntx][BGM_X3_BLOCK_BYTES]
};

template <int KTQ, int NTX, int NH, int WAVES, int PREC = 0, bool X4 = false>
__global__ __launch_bounds__(64 * WAVES) void bgm_hmc_kernel(BgmHmcKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BgmMeta &m = a.m;
  lds_fill(lds, a.blob, m.lds_resident);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  using HS = typename bgm_stream_of<PREC, WAVES, X4>::type;
  HS hs;
  if constexpr (PREC >= 1) hs.begin(a.hx3, m, lds);
  else if constexpr (NTX == 0) hs.begin(a.blob, m, lds);
  const long long n = a.n, n_tiles = (n + 15) / 16, passes = bgm_block_passes(n_tiles, WAVES);
  const float eps = *a.step;
  for (long long ps = 0; ps < passes; ++ps) {
    // Wide variant: tiles are dealt wave-major (wave w of every block before wave w + 1 of any), so the tiles of a partly filled last
    // round spread over ALL blocks as whole waves; a wave without a tile only keeps the block's head stream moving (no matrix work).
    // The round then costs what its ceil(active waves / 4) waves per SIMD cost instead of a full round on some CUs.
    long long tile = NTX == 0 ? (ps * WAVES + wave) * gridDim.x + blockIdx.x : (ps * gridDim.x + blockIdx.x) * WAVES + wave;
    const bool tile_ok = tile < n_tiles;
    if (NTX > 0 && !tile_ok) break;
    if constexpr (NTX == 0) {
      if (!tile_ok) {
        const int evals = (a.init ? 1 : 0) + a.n_iters * a.n_leapfrog;
        const int n_steps = PREC == 0 ? m.ntx : (m.ntx + BGM_X3_STEP - 1) / BGM_X3_STEP + (PREC == 2 ? 2 * ((NH + BGM_X3_STEP - 1) / BGM_X3_STEP) : 0);      // (steps of the stream per evaluation)
        for (int e = 0; e < evals; ++e)
          for (int tx = 0; tx < n_steps; ++tx) { hs.fetch(tx + 1 < n_steps ? tx + 1 : 0); hs.commit(); }
        continue;
      }
    }
    tile = tile_ok ? tile : n_tiles - 1;
    long long row = tile * 16 + j;
    const bool ok = tile_ok && row < n;
    row = row < n ? row : n - 1;
    const unsigned rowid = (unsigned)(a.row_base + row);
    BgmX<NTX> xr;
    f32x4 z[KTQ], gr[KTQ];
    bgm_load_x<NTX>(a.x, n, m.p, row, g, xr);
    if constexpr (PREC >= 1) hs.x_valid = false;       // (a new row: nothing of it has been requested ahead)
    float lp;
    if (a.init) {   // initial_state ~ N(0,1)  (bgm/base.py:778), RNG tag 0
#pragma unroll
      for (int t = 0; t < KTQ; ++t) {
        const f32x4 e = box_muller4(philox4x32_10(rowid, 0u, (unsigned)(g + 4 * t), TAG_INIT, a.k0, a.k1));
#pragma unroll
        for (int r = 0; r < 4; ++r) z[t][r] = (16 * t + 4 * r + g < m.q) ? e[r] : 0.0f;
      }
      bgm_logp_grad<KTQ, NTX, NH, true, PREC, HS>(lds, m, j, g, z, xr, hs, lp, gr);
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
          mom[t][r] = fmaf(0.5f * eps, gr[t][r], pm);   // first half kick
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
        bgm_logp_grad<KTQ, NTX, NH, true, PREC, HS>(lds, m, j, g, zc, xr, hs, lpc, gc, PREC == 0 || l == a.n_leapfrog - 1);
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
      // per-iteration statistics for SimpleStepSizeAdaptation and the acceptance report
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

// SimpleStepSizeAdaptation (target 0.75, rate 0.01 by default): one scalar step for all chains,
// multiplied / divided by (1 + rate) according to the mean acceptance probability of iteration `it`
// (mean of exp(min(0, log_accept_ratio)) == exp(reduce_logmeanexp)).
static __global__ void bgm_hmc_adapt_kernel(float *step, const double *acc_prob_sum, int it, double n_chains, float target,
                                     float rate) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double mean = acc_prob_sum[it] / n_chains;
    *step = (mean > (double)target) ? *step * (1.0f + rate) : *step / (1.0f + rate);
  }
}

// predict_on_posteriors (bgm/base.py:511-525): x ~ N(mu(z_d), sigma^2(z_d)) for every retained draw d.
//   full  [n_draws x n x p]                      (return_samples=True), or NULL
//   cells [(row * k_slots + slot) * n_draws + d]  for features with slot[row*p + c] >= 0, or NULL
struct BgmPredKArgs {
  const float *blob;
  const float *draws;       // [n_draws x n x q]
  long long n, row_base;
  int n_draws, burn_in, k_slots;
  const int *slot;          // [n x p] or NULL
  float *cells, *full, *var_full;   // var_full [n_draws x n x p]: sigma^2 of every cell, or NULL
  int add_noise;                     // 0: x = mu (use_x_sd=False paths)
  unsigned k0, k1;
  BgmMeta m;
};

template <int KTQ, int NTX, int NH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void bgm_predict_kernel(BgmPredKArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BgmMeta &m = a.m;
  lds_fill(lds, a.blob, m.lds_resident);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  BgmHeadStream hs;
  if constexpr (NTX == 0) hs.begin(a.blob, m, lds);
  const long long n = a.n, n_tiles = (n + 15) / 16;
  const long long work = n_tiles * a.n_draws, passes = bgm_block_passes(work, WAVES);
  for (long long ps = 0; ps < passes; ++ps) {
    BGM_NO_HOIST();
    long long w = (ps * gridDim.x + blockIdx.x) * WAVES + wave;
    const bool w_ok = w < work;
    if (NTX > 0 && !w_ok) break;
    w = w_ok ? w : work - 1;
    const long long tile = w / a.n_draws;
    const int d = (int)(w - tile * a.n_draws);
    long long row = tile * 16 + j;
    const bool ok = w_ok && row < n;
    row = row < n ? row : n - 1;
    const unsigned rowid = (unsigned)(a.row_base + row);
    f32x4 z[KTQ];
    bgm_load_z<KTQ>(a.draws + (long long)d * n * m.q, m.q, row, g, z);
    f32x4 h[4], h2[4];
    bias17<4>(lds + m.b1, g, h);
    fwd17<KTQ, 4>(lds + m.w1, j, g, z, h);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[t][r] = lrelu(h[t][r]);
    for (int l = 1; l < NH; ++l) {
      BGM_NO_HOIST();
      bias17<4>(lds + m.bh + (l - 1) * 64, g, h2);
      fwd17<4, 4>(lds + m.wh + (l - 1) * (4 * 64 * 17), j, g, h, h2);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[t][r] = lrelu(h2[t][r]);
    }
    auto head = [&](int tx, const float *wl) {
      f32x4 ms[2];
      ms[0] = *reinterpret_cast<const f32x4 *>(lds + m.bhd + 16 * tx + 4 * g);
      ms[1] = *reinterpret_cast<const f32x4 *>(lds + m.bhd + 16 * (m.ntx + tx) + 4 * g);
      heads_fwd17(wl, j, g, h, ms);
      // reparameterize (networks/base.py:113-117): noise = Philox tag 6, call (16 tx + 4 g)/4, outputs r
      const f32x4 e = box_muller4(philox4x32_10(rowid, (unsigned)(a.burn_in + d), (unsigned)(4 * tx + g), TAG_XNOISE, a.k0, a.k1));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * tx + 4 * g + r;
        if (ok && c < m.p) {
          const float s2 = softplus_f(ms[1][r]) + BGM_EPS;
          const float xp = a.add_noise ? fmaf(__builtin_sqrtf(s2), e[r], ms[0][r]) : ms[0][r];
          if (a.full) a.full[((long long)d * n + row) * m.p + c] = xp;
          if (a.var_full) a.var_full[((long long)d * n + row) * m.p + c] = s2;
          if (a.cells) {
            const int sl = a.slot[row * (long long)m.p + c];
            if (sl >= 0) a.cells[(row * (long long)a.k_slots + sl) * a.n_draws + d] = xp;
          }
        }
      }
    };
    if constexpr (NTX > 0) {
#pragma unroll
      for (int tx = 0; tx < NTX; ++tx) head(tx, lds + m.whd + tx * BGM_PAIR);
    } else {
#pragma unroll 1
      for (int tx = 0; tx < m.ntx; ++tx) {
        BGM_NO_HOIST();
        hs.fetch(tx + 1 < m.ntx ? tx + 1 : 0);
        head(tx, hs.tile());
        hs.commit();
      }
    }
  }
}
