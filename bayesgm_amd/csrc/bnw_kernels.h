// bnw_kernels.h -- posterior sampling, causal effects and evaluation of CausalBGM with Bayesian networks of ANY hidden width on gfx950:
// inference-mode input normalisation (params['bnn_norm'] = "fixed") or, as the reference is written (networks/bnn.py:25-27), the
// statistics of the batch a call sees = a block of `bs` rows of predict / the whole panel of evaluate ("batch": bnw_stats_kernel sums
// the columns of a block's states and proposals in fp64, every tile of the block normalises with them).
//
// replaces (src/bayesgm/models/causalbgm/base.py, use_bnn branches; networks/bnn.py:4-38 with any nb_units):
//   get_log_posterior :765-817, metropolis_hastings_sampler :820-904   -> bnw_noise_kernel + bnw_rows_kernel (modes 0 / 1)
//   infer_from_latent_posterior :671-763                               -> bnw_effects_kernel
//   evaluate :534-570, data_z = e_net(data_v) :479                     -> bnw_rows_kernel (modes 2 / 3) + bnw_effects_kernel
// The fast families (bnf_kernels.h: default shapes; bnn_sample_kernels.h: hidden widths <= 64) hold every layer in LDS fragments; this
// one is the correctness path for everything wider: a workgroup owns a tile of up to 64 rows of ONE block and walks the nets with the
// Flipout routines of the minibatch steps (bnn_kernels.h: activations in a per-workgroup global workspace, both products of a layer as
// one MFMA pass over 16 x 16 tiles).  With inference-mode normalisation the rows of a block share only the perturbation
// dW = sigma * eps of a (block, call), produced once per launch for all tiles (bnw_noise_kernel); the sign words are keyed by the row's
// position inside its block.  Noise specification: oracle/bnn.py (log_posterior_blocks, mh_iteration, effects_draw, evaluate).
#pragma once
#include <algorithm>

#include "bnn_kernels.h"

#define BNW_RT 64                       // rows of a workgroup tile
#ifndef BNW_WAVES_PER_EU
#define BNW_WAVES_PER_EU 2              // 2: one 512-thread workgroup per CU with 256 registers per lane; 4: two with 128
#endif

#ifdef BNW_PROF      // development: shader-clock cycles of wave 0 per phase of a net call, summed over workgroups (bnw_api.hip prints them)
__device__ unsigned long long bnw_prof_acc[8];
__device__ unsigned long long bnw_prof_last;      // (one value is enough for a relative split: every workgroup overwrites it in the same phase order)
#define BNW_T0() unsigned long long bnw_t_ = __builtin_amdgcn_s_memtime()
#define BNW_T(i) do { const unsigned long long t2_ = __builtin_amdgcn_s_memtime(); if (c.tid == 0) atomicAdd(&bnw_prof_acc[i], t2_ - bnw_t_); bnw_t_ = t2_; } while (0)
#define BNW_TG(i) do { const unsigned long long t2_ = __builtin_amdgcn_s_memtime(); if (tid == 0) atomicAdd(&bnw_prof_acc[i], t2_ - bnw_t_); bnw_t_ = t2_; } while (0)
#define BNW_TN(i, n) atomicAdd(&bnw_prof_acc[i], (unsigned long long)(n))
#else
#define BNW_TG(i) do {} while (0)
#define BNW_TN(i, n) do {} while (0)
#define BNW_T0() do {} while (0)
#define BNW_T(i) do {} while (0)
#endif

struct BnwNets;
struct BnwNets {
  BnnNet net[4];                        // g, e, f, h (BNN_* ids) with bn_fixed = 1
  const float *theta;
  const BnwNets *dev;                   // this struct in device memory (bnw_store_nets_kernel), what the row kernels read
  const float *locT;                    // posterior means of the nets of the sets, transposed, in the layout of one set (bnw_pack_kernel)
  int noff[4];                          // offset of net k's perturbation inside a set (-1: not in the sets of this launch)
  long long set_floats;
  int q, p, z0, z1, z2, binary;
  float sig2[3];                        // fixed sigma_v^2, sigma_x^2, sigma_y^2 (<= 0: the variance heads)
};

// dW of `n_calls` calls per block: set (blk * n_calls + c) = [net .. | net ..] with key of block block0 + blk, stream stream0 + c * stride
struct BnwNoiseArgs {
  BnwNets m;
  float *dw;
  int n_calls, block0;
  uint32_t k0, k1, stream0, stream_stride;
};
static __global__ __launch_bounds__(256) void bnw_noise_kernel(BnwNoiseArgs a) {
  const int set = blockIdx.y, blk = set / a.n_calls, call = set - blk * a.n_calls;
  const uint32_t stream = a.stream0 + (uint32_t)call * a.stream_stride, k1 = a.k1 + (uint32_t)(a.block0 + blk);
  float *base = a.dw + (long long)set * a.m.set_floats;
  for (int k = 0; k < 4; ++k) {
    if (a.m.noff[k] < 0) continue;
    const BnnNet &n = a.m.net[k];
    for (int l = 0; l < n.n_layers; ++l) {
      const int cnt = n.lin[l] * n.lout[l];
      const float *rho = a.m.theta + n.woff[l] + cnt;
      float *d = base + a.m.noff[k] + n.eoff[l];
      const int in = n.lin[l], out = n.lout[l];
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (cnt + 3) >> 2; i += gridDim.x * blockDim.x) {
        const f32x4 z = box_muller4(philox4x32_10((uint32_t)i, (uint32_t)l | ((uint32_t)n.net_id << 16), stream, BNN_TAG_EPS, a.k0, k1));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = 4 * i + u;      // element (row = input unit, column = output unit) of the layer's [in x out] kernel; stored transposed
          if (idx < cnt) { const int ki = idx / out, no = idx - ki * out; d[no * in + ki] = (BNN_SCALE_EPS + softplus_acc(rho[idx])) * z[u]; }
        }
      }
    }
  }
}

// posterior means of the nets of a set, transposed ([out x in] per layer), in the set's layout: once per launch sequence (theta is
// constant inside a sampling run)
static __global__ __launch_bounds__(256) void bnw_pack_kernel(BnwNets m, float *locT) {
  for (int k = 0; k < 4; ++k) {
    if (m.noff[k] < 0) continue;
    const BnnNet &n = m.net[k];
    for (int l = 0; l < n.n_layers; ++l) {
      const int in = n.lin[l], out = n.lout[l], cnt = in * out;
      const float *loc = m.theta + n.woff[l];
      float *d = locT + m.noff[k] + n.eoff[l];
      for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < cnt; idx += gridDim.x * blockDim.x) { const int ki = idx / out, no = idx - ki * out; d[no * in + ki] = loc[idx]; }
    }
  }
}

static __global__ __launch_bounds__(64) void bnw_store_nets_kernel(BnwNets m, BnwNets *dst) {
  const unsigned *src = reinterpret_cast<const unsigned *>(&m);
  unsigned *d = reinterpret_cast<unsigned *>(dst);
  for (int i = threadIdx.x; i < (int)(sizeof(BnwNets) / 4); i += 64) d[i] = src[i];
}

// ---- batch statistics of a block (params['bnn_norm'] = "batch") ---------------------------------------------------------------
// stats: double [2 parities][n_blocks][2 (0: proposal, 1: current state)][2 (sum, sum of squares)][64 columns] -- a launch accumulates
// into parity `par` and clears the other one for the next iteration; xstats: [n_blocks][2] sums of the treatment column (once per run).
// Same proposals as bnw_rows_kernel forms (bnw_normal), same layout as the narrow family's bns_propose_kernel.
struct BnwStatArgs {
  const float *z;
  long long n, row_base;
  int q, bs, wg_per_block, it, init;
  float q_sd;
  const float *q_sd_blocks;
  uint32_t k0, k1;
  double *stats;
  int n_blocks, par;
  const float *x; double *xstats;
};
__device__ __forceinline__ float bnw_normal(uint32_t row, uint32_t it, int f, uint32_t tag, uint32_t k0, uint32_t k1);
static __global__ __launch_bounds__(256) void bnw_stats_kernel(BnwStatArgs a) {
  __shared__ double red[4][4];
  const int blk = blockIdx.x / a.wg_per_block, wib = blockIdx.x - blk * a.wg_per_block;
  const long long r_in = (long long)wib * 256 + threadIdx.x, row = (long long)blk * a.bs + r_in;
  const bool valid = r_in < a.bs && row < a.n;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (blockIdx.x == 0) {
    double *clr = a.stats + (long long)(a.par ^ 1) * a.n_blocks * 256;
    for (long long i = threadIdx.x; i < (long long)a.n_blocks * 256; i += 256) clr[i] = 0.0;
  }
  const uint32_t rid = (uint32_t)(a.row_base + row);
  const float sd = a.q_sd_blocks ? a.q_sd_blocks[blk] : a.q_sd;
  double *st = a.stats + ((long long)a.par * a.n_blocks + blk) * 256;
  for (int f = 0; f < a.q; ++f) {
    float zc = 0.0f, zp = 0.0f;
    if (valid) {
      zc = a.init ? bnw_normal(rid, 0u, f, TAG_INIT, a.k0, a.k1) : a.z[row * a.q + f];
      zp = fmaf(sd, bnw_normal(rid, (uint32_t)a.it, f, TAG_PROP, a.k0, a.k1), zc);
    }
    double s[4] = {(double)zp, (double)zp * zp, (double)zc, (double)zc * zc};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_xor(s[k], off);
      if (lane == 0) red[wave][k] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(&st[threadIdx.x * 64 + f], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
  }
  if (a.xstats) {
    const double xv = valid ? (double)a.x[row] : 0.0;
    double xs = xv, xs2 = xv * xv;
    for (int off = 32; off > 0; off >>= 1) { xs += __shfl_xor(xs, off); xs2 += __shfl_xor(xs2, off); }
    if (lane == 0) { atomicAdd(&a.xstats[2 * blk], xs); atomicAdd(&a.xstats[2 * blk + 1], xs2); }
  }
}
// column sums / sums of squares of an [n x d] matrix: st[0 .. d) | st[d .. 2 d)   (evaluate: the encoder's input)
static __global__ __launch_bounds__(256) void bnw_colstats_kernel(const float *m, long long n, int d, double *st) {
  __shared__ double red[4][2];
  for (int col = blockIdx.y; col < d; col += gridDim.y) {
    double s = 0.0, s2 = 0.0;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < n; r += (long long)gridDim.x * 256) { const double x = (double)m[r * d + col]; s += x; s2 += x * x; }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); s2 += __shfl_xor(s2, off); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s; red[threadIdx.x >> 6][1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) { atomicAdd(&st[col], red[0][0] + red[1][0] + red[2][0] + red[3][0]); atomicAdd(&st[d + col], red[0][1] + red[1][1] + red[2][1] + red[3][1]); }
  }
}
// what a tile needs of its block's statistics (all NULL: inference-mode normalisation)
struct BnwStats {
  const double *st;      // [2 (sum, sum of squares)][64] of the latent columns of the state the call is made on
  const double *xst;     // [2] of the treatment column (NULL with `dose`)
  const double *vst;     // [2][p] of the covariates (encoder call)
  double cnt;            // rows of the block
  int dose;              // the treatment column is a constant (infer_from_latent_posterior: mean = the dose, variance 0)
  float dose_x;
};
// mean | variance of net `id`'s input columns -> ext [2 x in]   (column maps as bnw_inputs)
__device__ __forceinline__ void bnw_ext(const BnnCtx &c, const BnwNets &m, int id, const BnwStats &S, float *ext) {
  const int in = m.net[id].dims[0];
  for (int u = c.tid; u < in; u += BNN_THREADS) {
    double sm, sq;
    bool konst = false;
    if (id == BNN_E) { sm = S.vst[u]; sq = S.vst[in + u]; }
    else {
      int col = u;
      if (id == BNN_H) col = u < m.z0 ? u : u + m.z1;
      if (id == BNN_F && u >= m.z0 + m.z1) {
        if (S.dose) { konst = true; sm = sq = 0.0; }
        else { sm = S.xst[0]; sq = S.xst[1]; }
      } else { sm = S.st[col]; sq = S.st[64 + col]; }
    }
    if (konst) { ext[u] = S.dose_x; ext[in + u] = 0.0f; }
    else { const double mm = sm / S.cnt; ext[u] = (float)mm; ext[in + u] = (float)fmax(sq / S.cnt - mm * mm, 0.0); }
  }
  __syncthreads();
}

// ---- the layer products of this path ---------------------------------------------------------------------------------------------
// C1 = A1 W1, C2 = A2 W2 with A [M x K] row-major (the tile's activations and their sign-flipped copy in the workspace, M <= 64) and the
// weights TRANSPOSED, Wt [N x K] row-major (bnw_pack_kernel: posterior means; bnw_noise_kernel: the call's dW).  The workgroup walks K in
// chunks of 16 through a double-buffered LDS stage: per chunk the 512 threads fetch the chunk of both A operands (64 rows) and of both
// weight operands for a pass of 128 output columns -- 24 KiB, three 16-byte loads per thread, requested BEFORE the matrix instructions of
// the current chunk and written to the other half of the stage after them (one barrier per chunk).  Every element is fetched once per
// workgroup and pass instead of once per wave (the operands come from L2: the weights of all layers do not fit L1).  A wave owns a
// 32 x 32 block of the pass = 2 x 2 tiles of v_mfma_f32_16x16x4_f32 for each of the two products (eight independent accumulators) and
// reads its fragments as ds_read_b128 of four consecutive K: lane (j = lane & 15, g = lane >> 4) feeds A(m0 + j, k0 + 4 g + u),
// Wt(n0 + j, k0 + 4 g + u) in step u and receives C(m0 + 4 g + r, n0 + j).  Rows / columns beyond M / N are clamped duplicates whose
// results are not written; K beyond the last chunk is zero in the stage.
#define BNW_KC 16                                           // K of a chunk
#define BNW_NP 128                                          // output columns of a pass
#define BNW_STAGE_ROWS (2 * BNW_RT + 2 * BNW_NP)            // rows of one half of the stage: A1 | A2 | W1 | W2 chunks, BNW_KC floats each
#define BNW_STAGE_FLOATS (BNW_STAGE_ROWS * BNW_KC)
#define BNW_FETCH (BNW_STAGE_FLOATS / 4 / BNN_THREADS)      // 16-byte fetches per thread and chunk
// N <= 32 (the two-column heads of the outcome / treatment nets): a pass of 128 columns would leave six of the eight waves without a
// column.  Here the waves split K instead: wave (mb = wave >> 2, kg = wave & 3) takes the chunks kg, kg + 4, ... of the 32-row block mb
// straight from global memory (no stage, no barrier inside the loop); comb(c1, c2, row info, bit) = the two products joined (linear, so
// it may be applied to a wave's partial sums), the four partial results of a block meet in the stage and wave kg = 0 runs the epilogue
// with (sum, 0).
template <class RowInfo, class ColInfo, class Comb, class Epi>
__device__ __forceinline__ void bnw_gemm2_narrow(int tid, float *stage, const float *A1, const float *A2, int lda, const float *W1, const float *W2,
                                                 int M, int N, int K, RowInfo rowinfo, ColInfo colinfo, Comb comb, Epi epi) {
  const int lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6, mb = wave >> 2, kg = wave & 3;
  const bool vec = (lda & 3) == 0 && (K & 3) == 0 &&
                   ((((unsigned long long)A1 | (unsigned long long)A2 | (unsigned long long)W1 | (unsigned long long)W2) & 15ull) == 0);
  const int nc = (K + 15) >> 4, nt = (N + 15) >> 4;             // chunks of 16 K; column tiles (1 or 2)
  auto fetch = [&](const float *p, int k) -> f32x4 {
    if (vec && k + 3 < K) return *reinterpret_cast<const f32x4 *>(p + k);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (k + e < K) ? p[min(k + e, K - 1)] : 0.0f;
    return v;
  };
  int ra[2], cb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { ra[i] = min(32 * mb + 16 * i + j, M - 1) * lda; cb[i] = min(16 * i + j, N - 1) * K; }
  decltype(rowinfo(0, 0)) info[2][4];
  decltype(colinfo(0)) cinfo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    cinfo[i] = colinfo(min(16 * i + j, N - 1));
#pragma unroll
    for (int r = 0; r < 4; ++r) info[i][r] = rowinfo(min(32 * mb + 16 * i + 4 * g + r, M - 1), 0);
  }
  f32x4 c1[2][2], c2[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) { c1[i][i2] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; c2[i][i2] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
  for (int c = kg; c < nc; c += 4) {
    const int k = 16 * c + 4 * g;
    f32x4 a1[2], a2[2], b1[2], b2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { a1[i] = fetch(A1 + ra[i], k); a2[i] = fetch(A2 + ra[i], k); }
    b1[0] = fetch(W1 + cb[0], k); b2[0] = fetch(W2 + cb[0], k);
    if (nt > 1) { b1[1] = fetch(W1 + cb[1], k); b2[1] = fetch(W2 + cb[1], k); }
    BGM_NO_HOIST();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        c1[i][0] = BGM_MFMA(a1[i][u], b1[0][u], c1[i][0]);
        c2[i][0] = BGM_MFMA(a2[i][u], b2[0][u], c2[i][0]);
        if (nt > 1) { c1[i][1] = BGM_MFMA(a1[i][u], b1[1][u], c1[i][1]); c2[i][1] = BGM_MFMA(a2[i][u], b2[1][u], c2[i][1]); }
      }
  }
  // partial results: stage [wave][i][i2][lane] as float4
  f32x4 *part = reinterpret_cast<f32x4 *>(stage);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = comb(c1[i][i2][r], c2[i][i2][r], info[i][r], 16 * i2 + j);
      if (i2 < nt) part[((wave * 2 + i) * 2 + i2) * 64 + lane] = v;
    }
  __syncthreads();
  if (kg == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        if (i2 >= nt) continue;
        f32x4 v = part[((wave * 2 + i) * 2 + i2) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const f32x4 t = part[(((wave + w) * 2 + i) * 2 + i2) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += t[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 32 * mb + 16 * i + 4 * g + r, n = 16 * i2 + j;
          if (m < M && n < N) epi((unsigned)(m * N + n), v[r], 0.0f, info[i][r], cinfo[i2], 16 * i2 + j);
        }
      }
  }
  __syncthreads();      // the stage is free again
}

// rowinfo(m, n0) -> what the epilogue needs of output row m for the 32 columns from n0 (the sign words), colinfo(n) -> of column n (the
// bias): requested before the chunk loop -- a load inside the epilogue would wait behind the epilogue's own stores (one vmcnt counter);
// epi(m N + n, c1, c2, row info, column info, bit): bit = n - n0; comb: see bnw_gemm2_narrow.
// Schedule of a chunk (all eight waves meet at one barrier per chunk, so whatever a wave does outside its matrix instructions has to sit
// BETWEEN them, or the two waves of a SIMD leave the matrix pipe idle together): the fragments of chunk c are in registers when the
// iteration starts; step 0 of the four K steps -> the staged chunk c + 1 goes to the other half of the stage, the fetch of chunk c + 2 is
// requested -> step 1 -> barrier -> the fragments of chunk c + 1 are requested from LDS -> steps 2, 3.
struct BnwFrag { f32x4 a1[2], a2[2], b1[2], b2[2]; };
template <class RowInfo, class ColInfo, class Comb, class Epi>
__device__ __forceinline__ void bnw_gemm2(int tid, float *stage, const float *A1, const float *A2, int lda, const float *W1, const float *W2,
                                          int M, int N, int K, RowInfo rowinfo, ColInfo colinfo, Comb comb, Epi epi) {
  if (N <= 32) { bnw_gemm2_narrow(tid, stage, A1, A2, lda, W1, W2, M, N, K, rowinfo, colinfo, comb, epi); return; }
  static_assert(BNW_KC == 16, "the chunk schedule below is written for chunks of 16 K");
  constexpr int Q = BNW_KC / 4, RPS = BNN_THREADS / Q;       // 16-byte pieces of a stage row; stage rows per fetch slot
  const int lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
  const int mw = (wave >> 2) << 5, nw = (wave & 3) << 5;      // the wave's block inside the 64 x 128 pass
#ifndef BNW_SKEW
#define BNW_SKEW 1
#endif
  const bool late = BNW_SKEW ? (wave & BNW_SKEW) != 0 : false;
  const bool vec = (lda & 3) == 0 && (K & 3) == 0 &&
                   ((((unsigned long long)A1 | (unsigned long long)A2 | (unsigned long long)W1 | (unsigned long long)W2) & 15ull) == 0);
  const int nc = (K + BNW_KC - 1) / BNW_KC;
  const int fq = 4 * (tid % Q), fr = tid / Q;                 // fetch slot s of this thread: stage row s * RPS + fr, floats fq .. fq + 3
  auto fetch = [&](const float *p, int k) -> f32x4 {
    if (vec && k + 3 < K) return *reinterpret_cast<const f32x4 *>(p + k);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (k + e < K) ? p[min(k + e, K - 1)] : 0.0f;
    return v;
  };
  const int nc_fast = vec ? K / BNW_KC : 0;                   // chunks fetched as whole 16-byte pieces, no tests (wave-uniform)
  // fragment offsets of the lane inside a half of the stage
  int ar[2], br[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { ar[i] = (mw + 16 * i + j) * BNW_KC + 4 * g; br[i] = (2 * BNW_RT + nw + 16 * i + j) * BNW_KC + 4 * g; }
  auto frags = [&](BnwFrag &f, const float *h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f.a1[i] = *reinterpret_cast<const f32x4 *>(h + ar[i]); f.a2[i] = *reinterpret_cast<const f32x4 *>(h + BNW_RT * BNW_KC + ar[i]);
      f.b1[i] = *reinterpret_cast<const f32x4 *>(h + br[i]); f.b2[i] = *reinterpret_cast<const f32x4 *>(h + BNW_NP * BNW_KC + br[i]);
    }
  };
  for (int np = 0; np < N; np += BNW_NP) {
    const float *fp[BNW_FETCH];
#pragma unroll
    for (int sl = 0; sl < BNW_FETCH; ++sl) {
      const int row = sl * RPS + fr;                           // stage row: A1 [0, 64) | A2 [64, 128) | W1 [128, 256) | W2 [256, 384)
      if (row < BNW_RT) fp[sl] = A1 + min(row, M - 1) * lda;
      else if (row < 2 * BNW_RT) fp[sl] = A2 + min(row - BNW_RT, M - 1) * lda;
      else if (row < 2 * BNW_RT + BNW_NP) fp[sl] = W1 + min(np + row - 2 * BNW_RT, N - 1) * K;
      else fp[sl] = W2 + min(np + row - 2 * BNW_RT - BNW_NP, N - 1) * K;
    }
    float *sp = stage + fr * BNW_KC + fq;
    f32x4 c1[2][2], c2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) { c1[i][i2] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; c2[i][i2] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    const bool live = np + nw < N;                               // (wave-uniform) the wave's columns exist
    BNW_T0();
    f32x4 rg[BNW_FETCH];
#pragma unroll
    for (int sl = 0; sl < BNW_FETCH; ++sl) rg[sl] = fetch(fp[sl], fq);
    decltype(rowinfo(0, 0)) info[2][4];
    decltype(colinfo(0)) cinfo[2];
    if (live) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        cinfo[i] = colinfo(min(np + nw + 16 * i + j, N - 1));
#pragma unroll
        for (int r = 0; r < 4; ++r) info[i][r] = rowinfo(min(mw + 16 * i + 4 * g + r, M - 1), np + nw);
      }
    }
#pragma unroll
    for (int sl = 0; sl < BNW_FETCH; ++sl) *reinterpret_cast<f32x4 *>(sp + sl * RPS * BNW_KC) = rg[sl];
    if (nc > 1) {
#pragma unroll
      for (int sl = 0; sl < BNW_FETCH; ++sl) rg[sl] = fetch(fp[sl], BNW_KC + fq);
    }
    __syncthreads();
    BnwFrag f0, f1;
    if (live) frags(f0, stage);
    BNW_TG(3);
    auto step = [&](const BnwFrag &f, int u) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          c1[i][i2] = BGM_MFMA(f.a1[i][u], f.b1[i2][u], c1[i][i2]);
          c2[i][i2] = BGM_MFMA(f.a2[i][u], f.b2[i2][u], c2[i][i2]);
        }
    };
    auto chunk = [&](int c, const BnwFrag &cur, BnwFrag &nxt) {
      float *other = stage + ((c + 1) & 1) * BNW_STAGE_FLOATS;
      auto put = [&]() {
      if (c + 1 < nc) {
#pragma unroll
        for (int sl = 0; sl < BNW_FETCH; ++sl) *reinterpret_cast<f32x4 *>(other + (sp - stage) + sl * RPS * BNW_KC) = rg[sl];
        if (c + 2 < nc_fast) {
          const int k = (c + 2) * BNW_KC + fq;
#pragma unroll
          for (int sl = 0; sl < BNW_FETCH; ++sl) rg[sl] = *reinterpret_cast<const f32x4 *>(fp[sl] + k);
        } else if (c + 2 < nc) {
          const int k = (c + 2) * BNW_KC + fq;
#pragma unroll
          for (int sl = 0; sl < BNW_FETCH; ++sl) rg[sl] = fetch(fp[sl], k);
        }
      }
      };
      // the two waves of a SIMD (w, w + 4) place their stage writes / fragment reads at different K steps, so that one of them has
      // matrix instructions to issue while the other one is in its memory instructions
      if (late) put();
      BGM_NO_HOIST();
      if (live) step(cur, 0);
      BGM_NO_HOIST();
      if (!late) put();
      BGM_NO_HOIST();
      if (live) step(cur, 1);
      __syncthreads();
      if (!late && live && c + 1 < nc) frags(nxt, other);
      BGM_NO_HOIST();
      if (live) step(cur, 2);
      BGM_NO_HOIST();
      if (late && live && c + 1 < nc) frags(nxt, other);
      BGM_NO_HOIST();
      if (live) step(cur, 3);
    };
    int c = 0;
    for (; c + 2 <= nc; c += 2) { chunk(c, f0, f1); chunk(c + 1, f1, f0); }
    if (c < nc) chunk(c, f0, f1);
    BNW_TG(4);
    if (tid == 0) { BNW_TN(6, nc); BNW_TN(7, 1); }
    // every load of this pass has returned long ago; saying so here keeps hipcc from placing s_waitcnt vmcnt(0) in front of each
    // element of the epilogue (registers of the loop's fetches are reused there), where it waits for the previous element's STORES
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    if (live) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = mw + 16 * i + 4 * g + r, n = np + nw + 16 * i2 + j;
            if (m < M && n < N) epi((unsigned)(m * N + n), c1[i][i2][r], c2[i][i2][r], info[i][r], cinfo[i2], 16 * i2 + j);
          }
    }
    BNW_TG(5);
  }
}

// forward of the Flipout stack as bnn_layers_fwd (bnn_kernels.h), on the transposed weights: locT = the posterior means of the net in the
// layout of a perturbation set (layer l at eoff[l], [out x in]), k.dW = the call's perturbation in the same layout
__device__ __forceinline__ void bnw_layers_fwd(const BnnCtx &c, float *stage, const float *theta, const float *locT, const BnnNet &n, const BnnCache &k, int B) {
  const int L = n.n_layers;
  for (int l = 0; l < L; ++l) {
    const int in = n.lin[l], out = n.lout[l];
    const float *bias = theta + n.woff[l] + 2 * in * out;
    const float *h = k.H + (long long)B * n.hin[l], *hs = k.HS + (long long)B * n.hsin[l];
    float *y = k.H + (long long)B * n.hoff[l + 1], *ys = k.HS + (long long)B * n.hoff[l + 1];
    const bool last = (l == L - 1) || (n.heads && l == L - 2);
    const bool feeds_heads = n.heads && l == L - 3;
    float *ys2 = k.HS + (long long)B * n.hsin[L - 1];
    const int so = n.sout_w[l], si = last ? 0 : n.sin_w[l + 1], si2 = n.sin_w[L - 1];
    // the wave's 32 columns start at a multiple of 32: one word of the row's s_out string and one of the next layer's s_in string
    // (and of the variance head's, with heads) hold the bits of all of them
    struct Words { uint32_t so, si, si2; };
    bnw_gemm2(c.tid, stage, h, hs, in, locT + n.eoff[l], k.dW + n.eoff[l], B, out, in,
              [&](int m, int n0) {
                const uint32_t *sg = k.sg + (long long)m * n.swords + (n0 >> 5);
                Words w;
                w.so = sg[so]; w.si = last ? 0u : sg[si]; w.si2 = feeds_heads ? sg[si2] : 0u;
                return w;
              },
              [&](int o) { return bias[o]; },
              [&](float c1, float c2, const Words &w, int bit) { return c1 + (((w.so >> bit) & 1u) ? -c2 : c2); },
              [&](unsigned at, float c1, float c2, const Words &w, float bo, int bit) {      // at = m * out + o
                float v = c1 + bo + (((w.so >> bit) & 1u) ? -c2 : c2);
                if (!last) v = fmaxf(v, BNN_LEAK * v);
                y[at] = v;
                if (!last) ys[at] = ((w.si >> bit) & 1u) ? -v : v;
                if (feeds_heads) ys2[at] = ((w.si2 >> bit) & 1u) ? -v : v;
              });
    if (!(n.heads && l == L - 2)) __syncthreads();
  }
}

// call cache without private perturbation arrays (the call's dW is shared by all tiles of the block)
__device__ __forceinline__ void bnw_cache(const BnnNet &n, int B, float *p, BnnCache &k, const float *input, const float *dw) {
  auto take = [&](long long cnt) { float *r = p; p += (cnt + 3) & ~3LL; return r; };
  k.x = input;
  k.xhat = take((long long)B * n.dims[0]);
  k.inv = take(n.dims[0]);
  k.mu = nullptr;
  k.H = take((long long)B * n.hoff[n.n_layers + 1]);
  k.HS = take((long long)B * n.hs_total);
  k.eps = nullptr;
  k.dW = const_cast<float *>(dw);
  k.sg = (uint32_t *)take((long long)B * n.swords);
}
inline size_t bnw_cache_floats(const BnnNet &n, int B) {
  return (size_t)B * n.dims[0] + n.dims[0] + (size_t)B * (n.hoff[n.n_layers + 1] + n.hs_total) + (size_t)B * n.swords + 64;
}
// per-workgroup workspace: two latent tiles, the f / h inputs, the call cache
inline size_t bnw_ws_floats(const BnwNets &m) {
  size_t c = 0;
  for (int k = 0; k < 4; ++k) c = std::max(c, bnw_cache_floats(m.net[k], BNW_RT));
  int in_max = 0;
  for (int k = 0; k < 4; ++k) in_max = std::max(in_max, m.net[k].dims[0]);
  return (size_t)BNW_RT * (2 * (size_t)m.q + m.net[BNN_F].dims[0] + m.net[BNN_H].dims[0] + 8) + 2 * (size_t)in_max + 8 + c + 256;
}
struct BnwWs { float *zp, *zc, *fin, *hin, *ext, *cache, *stage; };      // (stage: the LDS stage of bnw_gemm2)
__device__ __forceinline__ void bnw_take(float *wp, const BnwNets &m, BnwWs &w) {
  auto take = [&](long long cnt) { float *r = wp; wp += (cnt + 3) & ~3LL; return r; };
  w.zp = take((long long)BNW_RT * m.q); w.zc = take((long long)BNW_RT * m.q);
  w.fin = take((long long)BNW_RT * m.net[BNN_F].dims[0]); w.hin = take((long long)BNW_RT * m.net[BNN_H].dims[0]);
  int in_max = 0;
  for (int k = 0; k < 4; ++k) in_max = max(in_max, m.net[k].dims[0]);
  w.ext = take(2LL * in_max);
  w.cache = wp;
}

// feature f of the row's normals(row, it, q, tag): Philox call (f & 3) + 4 (f >> 4), output (f >> 2) & 3   (oracle/rng.py normals)
__device__ __forceinline__ float bnw_normal(uint32_t row, uint32_t it, int f, uint32_t tag, uint32_t k0, uint32_t k1) {
  const f32x4 e = box_muller4(philox4x32_10(row, it, (uint32_t)((f & 3) + 4 * (f >> 4)), tag, k0, k1));
  const int r = (f >> 2) & 3;
  return r == 0 ? e[0] : (r == 1 ? e[1] : (r == 2 ? e[2] : e[3]));
}

// one call of net `id` on the tile's B rows (input [B x in] in the workspace or the panel); returns its output [B x wo]
__device__ __forceinline__ const float *bnw_call(const BnnCtx &c, const BnwNets &m, int id, const BnwWs &w, const float *input, int B,
                                                 const float *set, uint32_t k0, uint32_t k1, uint32_t stream, uint32_t row0,
                                                 const BnwStats *S = nullptr) {
  BnnCache k;
  bnw_cache(m.net[id], B, w.cache, k, input, set + m.noff[id]);
  BNW_T0();
  if (S) { bnw_ext(c, m, id, *S, w.ext); k.ext = w.ext; }
  const BnnNet &n = m.net[id];
  bnn_noise(c, m.theta, n, k, B, k0, k1, stream, row0, true);      // (the steps of bnn_fwd)
  __syncthreads();
  BNW_T(0);
  bnn_bn_fwd(c, m.theta, n, k, B);
  __syncthreads();
  BNW_T(1);
  bnw_layers_fwd(c, w.stage, m.theta, m.locT + m.noff[id], n, k, B);
  const float *o = k.H + (long long)B * n.hoff[n.heads ? n.n_layers - 1 : n.n_layers];
  __syncthreads();
  BNW_T(2);
  return o;
}
__device__ __forceinline__ void bnw_inputs(const BnnCtx &c, const BnwNets &m, const BnwWs &w, const float *zs, const float *xrow, int B) {
  const int q = m.q, nf = m.net[BNN_F].dims[0], nh = m.net[BNN_H].dims[0], zf = m.z0 + m.z1;
  for (int i = c.tid; i < B * nf; i += BNN_THREADS) { const int b = i / nf, j = i - b * nf; w.fin[i] = j < zf ? zs[b * q + j] : xrow[b]; }
  for (int i = c.tid; i < B * nh; i += BNN_THREADS) { const int b = i / nh, j = i - b * nh; w.hin[i] = j < m.z0 ? zs[b * q + j] : zs[b * q + m.z1 + j]; }
  __syncthreads();
}

// log p(z | x, y, v) + const of the tile's rows under the calls (set, stream): lp[b]   (base.py:765-817)
__device__ __forceinline__ void bnw_logp(const BnnCtx &c, const BnwNets &m, const BnwWs &w, const float *zs, const float *x, const float *y,
                                         const float *v, int B, const float *set, uint32_t k0, uint32_t k1, uint32_t stream, uint32_t row0,
                                         float *ssq, float *lp, const BnwStats *S = nullptr, const float *prior = nullptr) {
  // prior: the rows' conditional latent prior [B][q + 2] = mu [q], 1 / sigma^2, (q / 2) log sigma^2 (bprior_api.hip) or NULL = N(0, I)
  const int p = m.p, q = m.q;
  {
    const float *o = bnw_call(c, m, BNN_G, w, zs, B, set, k0, k1, stream, row0, S);
    const int wo = p + 1;
    bnn_row_ssq(c, v, o, B, p, wo, ssq);
    __syncthreads();
    for (int b = c.tid; b < B; b += BNN_THREADS) {
      const float s2 = m.sig2[0] > 0.0f ? m.sig2[0] : softplus_acc(o[b * wo + p]) + BGM_EPS;
      float zz = 0.0f, lc = 0.0f;
      if (prior) {
        const float *pr = prior + (long long)b * (q + 2);
        for (int j = 0; j < q; ++j) { const float d = zs[b * q + j] - pr[j]; zz = fmaf(d, d, zz); }
        zz *= pr[q]; lc = pr[q + 1];
      } else
      for (int j = 0; j < q; ++j) zz = fmaf(zs[b * q + j], zs[b * q + j], zz);
      lp[b] = -(ssq[b] / (2.0f * s2) + (float)p * logf(s2) * 0.5f) - 0.5f * zz - lc;
    }
    __syncthreads();
  }
  bnw_inputs(c, m, w, zs, x, B);
  {
    const float *o = bnw_call(c, m, BNN_H, w, w.hin, B, set, k0, k1, stream, row0, S);
    for (int b = c.tid; b < B; b += BNN_THREADS) {
      const float l = o[2 * b];
      if (m.binary) lp[b] -= fmaxf(l, 0.0f) - l * x[b] + log1pf(expf(-fabsf(l)));
      else { const float s2 = m.sig2[1] > 0.0f ? m.sig2[1] : softplus_acc(o[2 * b + 1]) + BGM_EPS, d = x[b] - l; lp[b] -= d * d / (2.0f * s2) + logf(s2) * 0.5f; }
    }
    __syncthreads();
  }
  {
    const float *o = bnw_call(c, m, BNN_F, w, w.fin, B, set, k0, k1, stream, row0, S);
    for (int b = c.tid; b < B; b += BNN_THREADS) {
      const float s2 = m.sig2[2] > 0.0f ? m.sig2[2] : softplus_acc(o[2 * b + 1]) + BGM_EPS, d = y[b] - o[2 * b];
      lp[b] -= d * d / (2.0f * s2) + logf(s2) * 0.5f;
    }
    __syncthreads();
  }
}

struct BnwRowsArgs {
  BnwNets m;                            // host side only: the kernels read the copy at mp (bnw_store_nets_kernel) -- indexing the nets'
  const BnwNets *mp;                    // layer tables inside a by-value argument makes the compiler copy the whole struct to scratch
  const float *dw;
  int n_calls;                          // sets per block
  const float *x, *y, *v;
  float *z;                             // [n x q]: state (modes 0, 1, 2), written by mode 3
  long long n, row_base;
  int bs, block0, tiles_per_block, n_items;
  int mode;                             // 0: log posterior of z -> out;  1: MH iteration `it`;  2: reconstruction sums;  3: z = e(v)
  int it, init;
  float q_sd;
  const float *q_sd_blocks;
  uint32_t k0, k1, stream0;
  float *out;                           // mode 0: [n]
  unsigned *acc_count, *acc_blocks;     // mode 1 (optional)
  double *sums;                         // mode 2: [3] += sums over rows of |v - v^|^2, (x - x^)^2, (y - y^)^2
  float *ws;
  long long ws_stride;
  // batch statistics (NULL: inference-mode normalisation): this launch's parity of bnw_stats_kernel's sums [n_blocks][2][2][64],
  // the treatment column's [n_blocks][2], the covariates' [2][p] (mode 3)
  const double *stats, *xstats, *vstats;
  // conditional latent prior (IdentifiableCausalBGM, bprior_kernels.h): [n_states][n][q + 2] per-row tables of the calls' own noisy
  // evaluations of the prior net (state 0: the proposal's / mode 0's, state 1: the current state's), or NULL
  const float *prior;
  long long prior_stride;
};

static __global__ __launch_bounds__(BNN_THREADS) __attribute__((amdgpu_waves_per_eu(BNW_WAVES_PER_EU, BNW_WAVES_PER_EU))) void bnw_rows_kernel(BnwRowsArgs a) {
  __shared__ float red[32];
  extern __shared__ __attribute__((aligned(16))) float bnw_stage[];      // 2 * BNW_STAGE_FLOATS (dynamic: bnw_api.hip)
  __shared__ float ssq[BNW_RT], lpp[BNW_RT], lpc[BNW_RT];
  __shared__ unsigned nacc_s;
  BnnCtx c{(int)threadIdx.x, red};
  const BnwNets &m = *a.mp;
  const int q = m.q, p = m.p;
  BnwWs w;
  bnw_take(a.ws + (long long)blockIdx.x * a.ws_stride, m, w);
  w.stage = bnw_stage;
  for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
    const int blk = item / a.tiles_per_block, t = item - blk * a.tiles_per_block;
    const long long blk_lo = (long long)blk * a.bs;
    const int blk_n = (int)min((long long)a.bs, a.n - blk_lo), rib0 = t * BNW_RT;
    if (rib0 >= blk_n) continue;
    const int B = min(BNW_RT, blk_n - rib0);
    const long long r0 = blk_lo + rib0;
    const uint32_t k1b = a.k1 + (uint32_t)(a.block0 + blk);
    const float *xr = a.x ? a.x + r0 : nullptr, *yr = a.y ? a.y + r0 : nullptr, *vr = a.v ? a.v + r0 * p : nullptr;
    const float *set0 = a.dw + (long long)blk * a.n_calls * m.set_floats;
    const bool batch = a.stats != nullptr || a.vstats != nullptr;
    BnwStats Sc{}, Sp{};                // statistics of the block's current states / proposals
    if (batch) {
      Sc.cnt = Sp.cnt = (double)blk_n;
      Sc.vst = Sp.vst = a.vstats;
      Sc.xst = Sp.xst = a.xstats ? a.xstats + 2 * blk : nullptr;
      if (a.stats) { Sp.st = a.stats + (long long)blk * 256; Sc.st = Sp.st + 128; }
    }
    const BnwStats *pSc = batch ? &Sc : nullptr, *pSp = batch ? &Sp : nullptr;
    if (a.mode == 3) {                  // data_z = e_net(data_v)
      const float *o = bnw_call(c, m, BNN_E, w, vr, B, set0, a.k0, k1b, a.stream0, (uint32_t)rib0, pSc);
      for (int i = c.tid; i < B * q; i += BNN_THREADS) a.z[r0 * q + i] = o[i];
      __syncthreads();
      continue;
    }
    {
      const float sd = a.q_sd_blocks ? a.q_sd_blocks[blk] : a.q_sd;
      for (int i = c.tid; i < B * q; i += BNN_THREADS) {
        const uint32_t rid = (uint32_t)(a.row_base + r0 + i / q);
        float zv;
        if (a.mode == 1 && a.init) { zv = bnw_normal(rid, 0u, i % q, TAG_INIT, a.k0, a.k1); a.z[r0 * q + i] = zv; }      // current_state ~ N(0, 1), base.py:842
        else zv = a.z[r0 * q + i];
        w.zc[i] = zv;
        if (a.mode == 1) w.zp[i] = fmaf(sd, bnw_normal(rid, (uint32_t)a.it, i % q, TAG_PROP, a.k0, a.k1), zv);
      }
    }
    __syncthreads();
    if (a.mode == 0) {
      bnw_logp(c, m, w, w.zc, xr, yr, vr, B, set0, a.k0, k1b, a.stream0, (uint32_t)rib0, ssq, lpc, pSc, a.prior ? a.prior + r0 * (q + 2) : nullptr);
      for (int b = c.tid; b < B; b += BNN_THREADS) a.out[r0 + b] = lpc[b];
      __syncthreads();
    } else if (a.mode == 1) {
      bnw_logp(c, m, w, w.zp, xr, yr, vr, B, set0, a.k0, k1b, 2u * (uint32_t)a.it, (uint32_t)rib0, ssq, lpp, pSp,
               a.prior ? a.prior + r0 * (q + 2) : nullptr);
      bnw_logp(c, m, w, w.zc, xr, yr, vr, B, set0 + m.set_floats, a.k0, k1b, 2u * (uint32_t)a.it + 1u, (uint32_t)rib0, ssq, lpc, pSc,
               a.prior ? a.prior + a.prior_stride + r0 * (q + 2) : nullptr);
      if (c.tid == 0) nacc_s = 0u;
      __syncthreads();
      for (int b = c.tid; b < B; b += BNN_THREADS) {
        const uint4 w4 = philox4x32_10((uint32_t)(a.row_base + r0 + b), (uint32_t)a.it >> 2, 0u, TAG_ACC, a.k0, a.k1);
        const int it = a.it;
        const unsigned wd = (it & 2) ? ((it & 1) ? w4.w : w4.z) : ((it & 1) ? w4.y : w4.x);
        if (u01_open(wd) < expf(fminf(lpp[b] - lpc[b], 0.0f))) {
          for (int j = 0; j < q; ++j) a.z[(r0 + b) * q + j] = w.zp[b * q + j];
          atomicAdd(&nacc_s, 1u);
        }
      }
      __syncthreads();
      if (c.tid == 0 && nacc_s) {
        if (a.acc_count) atomicAdd(a.acc_count, nacc_s);
        if (a.acc_blocks) atomicAdd(&a.acc_blocks[blk], nacc_s);
      }
      __syncthreads();
    } else {                            // mode 2: one call of g, h, f -> squared reconstruction errors (base.py:541-552)
      float sv = 0.0f, sx = 0.0f, sy = 0.0f;
      {
        const float *o = bnw_call(c, m, BNN_G, w, w.zc, B, set0, a.k0, k1b, a.stream0, (uint32_t)rib0, pSc);
        bnn_row_ssq(c, vr, o, B, p, p + 1, ssq);
        __syncthreads();
        for (int b = c.tid; b < B; b += BNN_THREADS) sv += ssq[b];
        __syncthreads();
      }
      bnw_inputs(c, m, w, w.zc, xr, B);
      {
        const float *o = bnw_call(c, m, BNN_H, w, w.hin, B, set0, a.k0, k1b, a.stream0, (uint32_t)rib0, pSc);
        for (int b = c.tid; b < B; b += BNN_THREADS) { const float l = o[2 * b], xp = m.binary ? sigmoid_f(l) : l, d = xr[b] - xp; sx = fmaf(d, d, sx); }
        __syncthreads();
      }
      {
        const float *o = bnw_call(c, m, BNN_F, w, w.fin, B, set0, a.k0, k1b, a.stream0, (uint32_t)rib0, pSc);
        for (int b = c.tid; b < B; b += BNN_THREADS) { const float d = yr[b] - o[2 * b]; sy = fmaf(d, d, sy); }
        __syncthreads();
      }
      sv = bnn_block_sum(c, sv); sx = bnn_block_sum(c, sx); sy = bnn_block_sum(c, sy);
      if (c.tid == 0) { atomicAdd(&a.sums[0], (double)sv); atomicAdd(&a.sums[1], (double)sx); atomicAdd(&a.sums[2], (double)sy); }
      __syncthreads();
    }
  }
}

// f-net at the treatment values xvals[0 .. n_doses) on one state per row (infer_from_latent_posterior, one kept draw per launch; the
// dose loops of evaluate).  Set of (block, dose k) = blk * n_doses + k, stream stream0 + k.
struct BnwEffArgs {
  BnwNets m;
  const BnwNets *mp;                    // (as BnwRowsArgs)
  const float *dw;
  const float *z;                       // [n x q]
  long long n, row_base;
  int bs, block0, tiles_per_block, n_items, n_doses;
  const float *xvals;
  uint32_t k0, k1, stream0, it_noise;
  int sample_y;
  double *sum_out; long long sum_stride;    // dose k: sum_out[k * sum_stride] += sum over rows (or NULL)
  float *ite_out; long long ite_stride;     // binary: ite_out[row * ite_stride] = y(dose 0) - y(dose 1) (or NULL)
  float *ws;
  long long ws_stride;
  const double *stats;                      // batch statistics of the states: [n_blocks][2][2][64], slot 1 (NULL: inference mode)
};
static __global__ __launch_bounds__(BNN_THREADS) __attribute__((amdgpu_waves_per_eu(BNW_WAVES_PER_EU, BNW_WAVES_PER_EU))) void bnw_effects_kernel(BnwEffArgs a) {
  __shared__ float red[32];
  extern __shared__ __attribute__((aligned(16))) float bnw_stage[];      // 2 * BNW_STAGE_FLOATS (dynamic: bnw_api.hip)
  __shared__ float y0[BNW_RT];
  BnnCtx c{(int)threadIdx.x, red};
  const BnwNets &m = *a.mp;
  const int q = m.q, nf = m.net[BNN_F].dims[0], zf = m.z0 + m.z1;
  BnwWs w;
  bnw_take(a.ws + (long long)blockIdx.x * a.ws_stride, m, w);
  w.stage = bnw_stage;
  for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
    const int blk = item / a.tiles_per_block, t = item - blk * a.tiles_per_block;
    const long long blk_lo = (long long)blk * a.bs;
    const int blk_n = (int)min((long long)a.bs, a.n - blk_lo), rib0 = t * BNW_RT;
    if (rib0 >= blk_n) continue;
    const int B = min(BNW_RT, blk_n - rib0);
    const long long r0 = blk_lo + rib0;
    const uint32_t k1b = a.k1 + (uint32_t)(a.block0 + blk);
    for (int k = 0; k < a.n_doses; ++k) {
      const float xv = a.xvals[k];
      for (int i = c.tid; i < B * nf; i += BNN_THREADS) { const int b = i / nf, j = i - b * nf; w.fin[i] = j < zf ? a.z[(r0 + b) * q + j] : xv; }
      __syncthreads();
      const float *set = a.dw + ((long long)blk * a.n_doses + k) * m.set_floats;
      BnwStats S{};
      if (a.stats) { S.st = a.stats + (long long)blk * 256 + 128; S.cnt = (double)blk_n; S.dose = 1; S.dose_x = xv; }
      const float *o = bnw_call(c, m, BNN_F, w, w.fin, B, set, a.k0, k1b, a.stream0 + (uint32_t)k, (uint32_t)rib0, a.stats ? &S : nullptr);
      float tot = 0.0f;
      for (int b = c.tid; b < B; b += BNN_THREADS) {
        float yk = o[2 * b];
        if (a.sample_y) {
          const float s2 = m.sig2[2] > 0.0f ? m.sig2[2] : softplus_acc(o[2 * b + 1]) + BGM_EPS;
          const f32x4 nz = box_muller4(philox4x32_10((uint32_t)(a.row_base + r0 + b), a.it_noise, (uint32_t)(k >> 2), TAG_YNOISE, a.k0, a.k1));
          const int e = k & 3;
          yk = fmaf(sqrtf(s2), e == 0 ? nz[0] : (e == 1 ? nz[1] : (e == 2 ? nz[2] : nz[3])), yk);
        }
        tot += yk;
        if (a.ite_out) {
          if (k == 0) y0[b] = yk;
          else if (k == 1) a.ite_out[(r0 + b) * a.ite_stride] = y0[b] - yk;
        }
      }
      if (a.sum_out) {
        tot = bnn_block_sum(c, tot);
        if (c.tid == 0) atomicAdd(&a.sum_out[(long long)k * a.sum_stride], (double)tot);
      }
      __syncthreads();
    }
  }
}
