// bnn_sample_kernels.h -- CausalBGM with Bayesian networks (use_bnn=True): large-batch forward kernels on gfx950
// (posterior sampling, causal effects, evaluation).
//
// replaces (src/bayesgm/models/causalbgm/base.py, use_bnn branches):
//   get_log_posterior :765-817 + metropolis_hastings_sampler :820-904  -> bns_propose_kernel, bns_noise_kernel, bns_mh_kernel
//   infer_from_latent_posterior :671-763                                -> bns_effects_kernel
//   evaluate :534-570                                                   -> bns_eval_* kernels
//
// With Bayesian nets every log-posterior evaluation normalises its input with the statistics of the block of rows it
// is given (bs rows of predict) and draws ONE weight perturbation per layer for the whole block.  That couples the
// chains of a block, and neither the current state's log-posterior nor the weights can be kept across iterations: the
// persistent weights-in-LDS design of causal_kernels.h does not apply.  Structure here, one MH iteration over ALL
// blocks of the panel in lock step (blocks are independent, so N / bs of them advance together):
//   bns_noise_kernel    dW = sigma * eps of every (block, call), written in MFMA fragment order
//   bns_propose_kernel  proposal z' = z + q_sd * N(0, 1); column sums of z and z' per block (fp64 atomics)
//   bns_mh_kernel       per workgroup 128 rows of one block: g, h, f forward for z' and z, accept / reject
// Per net the workgroup (4 waves, two workgroups per CU) runs layer-synchronously: the layer's loc and dW fragments are
// staged in LDS in chunks of 2 x 16 KB,
// every wave keeps the activations of its 2 x 16 rows in registers (swapped MFMA orientation: output units along M,
// rows along N, so accumulators are the next layer's B operands) and runs both GEMMs of the Flipout layer,
// y = loc^T h + s_out * (dW^T (s_in * h)) + b, with two accumulators per output tile.
#pragma once
#include <hip/hip_runtime.h>

#include "bnn_kernels.h"

#define BNS_WAVES 8
#define BNS_THREADS (64 * BNS_WAVES)
#define BNS_CHUNK 16      // (output tile, k tile) fragment pairs per staged chunk: 2 arrays x 16 KB
#define BNS_R 1
#define BNS_ROWS (BNS_WAVES * 16 * BNS_R)
#define BNS_MAXT 4        // hidden widths <= 64
#define BNS_SW 32         // sign words reserved per row in LDS
#define BNS_MAXK 208      // widest network input (padded)
#define BNS_MAXB 1024     // all biases of a net (each layer padded to whole tiles), staged in LDS

struct BnsNet {
  int n_layers, net_id, bn_fixed;
  int K[BNN_MAX_LAYERS + 1];
  int T[BNN_MAX_LAYERS], MT[BNN_MAX_LAYERS];   // input k-tiles / output tiles of layer l
  int foff[BNN_MAX_LAYERS + 1];                // fragments of layer l inside the net's block; foff[L] = block size
  int woff[BNN_MAX_LAYERS];                    // loc of layer l in theta (rho, bias follow as in BnnNet)
  int goff;                                    // gamma in theta (beta follows)
  int sin_w[BNN_MAX_LAYERS], sout_w[BNN_MAX_LAYERS], swords;
  int fbase;                                   // this net's block inside the packed loc / sigma arrays
  int dbase;                                   // ... inside one (block, call) perturbation set
};
inline void bns_from(const BnnNet &b, BnsNet &n) {
  n.n_layers = b.n_layers; n.net_id = b.net_id; n.goff = b.off; n.swords = b.swords; n.bn_fixed = b.bn_fixed;
  int f = 0;
  for (int l = 0; l <= b.n_layers; ++l) n.K[l] = b.dims[l];
  for (int l = 0; l < b.n_layers; ++l) {
    n.T[l] = (b.dims[l] + 15) / 16; n.MT[l] = (b.dims[l + 1] + 15) / 16;
    n.foff[l] = f; f += n.T[l] * n.MT[l] * 256;
    n.woff[l] = b.woff[l]; n.sin_w[l] = b.sin_w[l]; n.sout_w[l] = b.sout_w[l];
  }
  n.foff[b.n_layers] = f;
}

// Fragment order of a kernel [in x out]: float4 per (mt, t, lane): element r = W[k = 16 t + 4 g + r][o = 16 mt + j],
// lane = 16 g + j, zero outside the matrix.
__device__ __forceinline__ int bns_frag_pos(int k, int o, int T) {
  const int t = k >> 4, g = (k >> 2) & 3, r = k & 3, mt = o >> 4, j = o & 15;
  return (((mt * T + t) << 6) + (g << 4) + j) * 4 + r;
}

// loc and sigma = eps + softplus(rho) of every layer into fragment order (once per parameter change)
struct BnsPackArgs { BnsNet net[4]; const float *theta; float *lf, *sf; int n_nets; };
static __global__ void bns_pack_kernel(BnsPackArgs a) {
  const int k_ = blockIdx.y;
  const BnsNet &n = a.net[k_];
  for (int l = 0; l < n.n_layers; ++l) {
    const int in = n.K[l], out = n.K[l + 1], cnt = in * out;
    const float *loc = a.theta + n.woff[l], *rho = loc + cnt;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
      const int pos = n.fbase + n.foff[l] + bns_frag_pos(i / out, i % out, n.T[l]);
      // layers behind a LeakyReLU consume lrelu_s(x) = LeakyReLU(x) / 0.6 (one instruction, bgm_device.h): their loc and
      // sigma (hence every perturbation sigma * eps) carry the factor 0.6
      const float sc = l > 0 ? BGM_LRS_W : 1.0f;
      a.lf[pos] = sc * loc[i];
      a.sf[pos] = sc * (BNN_SCALE_EPS + softplus_acc(rho[i]));
    }
  }
}

// dW fragments of every (block, call): grid (chunks, n_sets); set s = block * n_calls + call has noise key
// (k0, k1 + block) and stream stream0 + call * stream_stride.
struct BnsNoiseArgs {
  BnsNet net[4];
  int n_nets, n_calls;
  const float *sf;
  float *dw;                 // [n_sets][set_floats]
  long long set_floats;
  uint32_t k0, k1, stream0, stream_stride;
  int block0;                // batch id of the first block (key offset)
};
static __global__ void bns_noise_kernel(BnsNoiseArgs a) {
  const int set = blockIdx.y, blk = set / a.n_calls, call = set - blk * a.n_calls;
  const uint32_t k1 = a.k1 + (uint32_t)(a.block0 + blk), stream = a.stream0 + (uint32_t)call * a.stream_stride;
  float *dw = a.dw + (long long)set * a.set_floats;
  for (int k_ = 0; k_ < a.n_nets; ++k_) {
    const BnsNet &n = a.net[k_];
    for (int l = 0; l < n.n_layers; ++l) {
      const int out = n.K[l + 1], cnt = n.K[l] * out;
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (cnt + 3) >> 2; i += gridDim.x * blockDim.x) {
        const f32x4 z = box_muller4(philox4x32_10((uint32_t)i, (uint32_t)l | ((uint32_t)n.net_id << 16), stream, BNN_TAG_EPS, a.k0, k1));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = 4 * i + u;
          if (idx < cnt) {
            const int pos = n.foff[l] + bns_frag_pos(idx / out, idx % out, n.T[l]);
            dw[n.dbase + pos] = a.sf[n.fbase + pos] * z[u];
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// proposal + column statistics.  One thread per row, 256 rows of ONE block per workgroup.
// stats: double [n_blocks][2 (0: proposal, 1: current)][2 (sum, sum of squares)][64]; this launch accumulates into
// parity `par` and clears parity par ^ 1 for the next iteration.
// ---------------------------------------------------------------------------------------------
struct BnsPropArgs {
  const float *z; float *zprop;
  long long n, row_base;
  int q, bs, wg_per_block, it, init;     // init: z itself is drawn here (iteration-0 state, TAG_INIT) before proposing
  float q_sd;
  const float *q_sd_blocks;              // optional per-block proposal scale
  uint32_t k0, k1;
  double *stats;                         // [2 parities][n_blocks][2][2][64]
  int n_blocks, par;
  float *z_init;                         // written when init
  const float *x; double *xstats;        // when xstats != NULL: column sums of x per block [n_blocks][2] (once per run)
};
__device__ __forceinline__ void bns_row_normals(uint32_t row, uint32_t it, int q, uint32_t tag, uint32_t k0, uint32_t k1, float *out) {
  const int n_s = (q + 3) >> 2;
  for (int sb = 0; sb < (n_s + 3) >> 2; ++sb)
    for (int g = 0; g < 4 && g < q; ++g) {
      const f32x4 e = box_muller4(philox4x32_10(row, it, (uint32_t)(g + 4 * sb), tag, k0, k1));
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int f = 16 * sb + 4 * r + g; if (f < q) out[f] = e[r]; }
    }
}
static __global__ __launch_bounds__(256) void bns_propose_kernel(BnsPropArgs a) {
  __shared__ double red[4][4][64];   // [wave][quantity][column]
  const int blk = blockIdx.x / a.wg_per_block, wib = blockIdx.x - blk * a.wg_per_block;
  const long long r_in = (long long)wib * 256 + threadIdx.x;
  const long long row = (long long)blk * a.bs + r_in;
  const bool valid = r_in < a.bs && row < a.n;
  const int q = a.q, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (blockIdx.x == 0) {
    double *clr = a.stats + (long long)(a.par ^ 1) * a.n_blocks * 256;
    for (long long i = threadIdx.x; i < (long long)a.n_blocks * 256; i += 256) clr[i] = 0.0;
  }
  float zc[64], e[64];
  if (valid) {
    const uint32_t rid = (uint32_t)(a.row_base + row);
    if (a.init) {
      bns_row_normals(rid, 0u, q, TAG_INIT, a.k0, a.k1, zc);
      for (int f = 0; f < q; ++f) a.z_init[row * q + f] = zc[f];
    } else {
      for (int f = 0; f < q; ++f) zc[f] = a.z[row * q + f];
    }
    bns_row_normals(rid, (uint32_t)a.it, q, TAG_PROP, a.k0, a.k1, e);
    const float sd = a.q_sd_blocks ? a.q_sd_blocks[blk] : a.q_sd;
    for (int f = 0; f < q; ++f) { e[f] = fmaf(sd, e[f], zc[f]); a.zprop[row * q + f] = e[f]; }
  }
  for (int f = 0; f < q; ++f) {
    double s[4] = {valid ? (double)e[f] : 0.0, valid ? (double)e[f] * e[f] : 0.0, valid ? (double)zc[f] : 0.0,
                   valid ? (double)zc[f] * zc[f] : 0.0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_xor(s[k], off);
      if (lane == 0) red[wave][k][f] = s[k];
    }
  }
  double xs = 0.0, xs2 = 0.0;
  if (a.xstats) {
    const double xv = valid ? (double)a.x[row] : 0.0;
    xs = xv; xs2 = xv * xv;
    for (int off = 32; off > 0; off >>= 1) { xs += __shfl_xor(xs, off); xs2 += __shfl_xor(xs2, off); }
  }
  __syncthreads();
  double *st = a.stats + ((long long)a.par * a.n_blocks + blk) * 256;
  for (int i = threadIdx.x; i < 4 * q; i += 256) {
    const int k = i / q, f = i - k * q;
    atomicAdd(&st[k * 64 + f], red[0][k][f] + red[1][k][f] + red[2][k][f] + red[3][k][f]);
  }
  if (a.xstats && lane == 0) { atomicAdd(&a.xstats[2 * blk], xs); atomicAdd(&a.xstats[2 * blk + 1], xs2); }
}

// ---------------------------------------------------------------------------------------------
// network forward of one workgroup's rows
// ---------------------------------------------------------------------------------------------
#ifdef BNS_PROF
#define BNS_T(c_, k) { BnsCtx &cc_ = const_cast<BnsCtx &>(c_); const unsigned long long t_ = clock64(); cc_.tp[k] += t_ - cc_.t_last; cc_.t_last = t_; }
#else
#define BNS_T(c_, k)
#endif
struct BnsCtx {
#ifdef BNS_PROF
  unsigned long long tp[6] = {0, 0, 0, 0, 0, 0}, t_last = 0;   // 0 prologue, 1 staging, 2 first layer, 3 middle layers, 4 last layer, 5 outside
#endif
  int tid, wave, lane, j, g;
  float *stage;        // LDS: loc fragments | dW fragments of the current layer
  uint32_t *sg;        // LDS: sign words [wave][rt][16 rows][BNS_SW]
  float *bn;           // LDS: scale[BNS_MAXK] | shift[BNS_MAXK] | bias of the current layer [BNS_MAXB]
};

// Workgroup id -> work item such that the workgroups of ONE block of rows (which share a perturbation set and the packed
// kernels' cache lines) run on ONE XCD: the dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs, so the
// ids congruent modulo 8 get a contiguous range of work items.  Returns -1 for the padding ids of the last round.
__device__ __forceinline__ int bns_xcd_item(int total) {
  const int per = (total + 7) >> 3;
  const int item = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  return ((int)(blockIdx.x >> 3) < per && item < total) ? item : -1;
}

__device__ __forceinline__ float bns_flip(float x, uint32_t w, int bit) {
  return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) ^ ((w >> bit) << 31));
}
__device__ __forceinline__ uint32_t bns_sgw(const BnsCtx &c, int rt, int word) {
  return c.sg[(((c.wave * BNS_R + rt) << 4) + c.j) * BNS_SW + word];
}

// stat(u) -> (mean, variance) of input column u of this call's batch; in(rt, u) -> raw input value of row (rt, j),
// column u (< K); epi(rt, mt, y): last layer's outputs, y[r] = unit 16 mt + 4 g + r of row (rt, j).
// rib0: index of the workgroup's first row inside its block (sign rows).
struct BnsNoPre { __device__ void operator()(int) const {} };
template <class Stat, class In, class Epi, class Pre = BnsNoPre>
__device__ __forceinline__ void bns_forward(const BnsCtx &c, const BnsNet &n, const float *theta, const float *lf, const float *dw,
                                            uint32_t k0, uint32_t k1, uint32_t stream, int rib0, Stat stat, In in, Epi epi,
                                            Pre pre = Pre()) {
  const int L = n.n_layers, j = c.j, g = c.g, lane = c.lane;
  BNS_T(c, 5);
  __syncthreads();    // previous users of bn / sg / stage are done
  {
    const float *gamma = theta + n.goff, *beta = gamma + n.K[0];
    for (int u = c.tid; u < 16 * n.T[0]; u += BNS_THREADS) {
      float sc = 0.0f, sh = 0.0f;
      if (u < n.K[0]) {
        float mean = 0.0f, var = 1.0f;
        if (!n.bn_fixed) stat(u, mean, var);
        sc = gamma[u] / sqrtf(var + BNN_BN_EPS);
        sh = beta[u] - mean * sc;
      }
      c.bn[u] = sc; c.bn[BNS_MAXK + u] = sh;
    }
    {   // all biases of the net (layer l at 16 * (MT[0] + ... + MT[l-1])): unconditional loads, nothing waits in a loop
      int bo = 0;
      for (int l = 0; l < L; ++l) {
        const float *bias = theta + n.woff[l] + 2 * n.K[l] * n.K[l + 1];
        const int M = n.K[l + 1];
        for (int o = c.tid; o < 16 * n.MT[l]; o += BNS_THREADS) c.bn[2 * BNS_MAXK + bo + o] = bias[min(o, M - 1)] * (o < M ? 1.0f : 0.0f);
        bo += 16 * n.MT[l];
      }
    }
    const int calls = n.swords >> 2;
#pragma unroll
    for (int rt = 0; rt < BNS_R; ++rt) {
      const uint32_t row = (uint32_t)(rib0 + ((c.wave * BNS_R + rt) << 4) + j);
      for (int cc = g; cc < calls; cc += 4) {
        const uint4 w = philox4x32_10(row, (uint32_t)cc | ((uint32_t)n.net_id << 16), stream, BNN_TAG_SIGN, k0, k1);
        uint32_t *dst = c.sg + (((c.wave * BNS_R + rt) << 4) + j) * BNS_SW + 4 * cc;
        dst[0] = w.x; dst[1] = w.y; dst[2] = w.z; dst[3] = w.w;
      }
    }
  }
  float h[BNS_R][BNS_MAXT][4];
  // first-layer inputs of narrow nets (one k-tile) are requested before the staging barriers
  float xin[BNS_R][4];
  if (n.T[0] == 1) {
#pragma unroll
    for (int rt = 0; rt < BNS_R; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) xin[rt][r] = in(rt, 4 * g + r);
  }
  // A net whose fragments fit one chunk (f, h) is staged whole: one barrier pair per call instead of one per layer.
  const bool whole = n.foff[L] <= BNS_CHUNK * 256;
  BNS_T(c, 0);
  if (whole) {
    const f32x4 *s1 = (const f32x4 *)lf, *s2 = (const f32x4 *)dw;
    f32x4 *d = (f32x4 *)c.stage;
    __syncthreads();
    {
      constexpr int PF = (BNS_CHUNK * 64 + BNS_THREADS - 1) / BNS_THREADS;
      const int cnt = n.foff[L] >> 2;
      f32x4 v1[PF], v2[PF];
#pragma unroll
      for (int u = 0; u < PF; ++u) { const int i = min(c.tid + u * BNS_THREADS, cnt - 1); v1[u] = s1[i]; v2[u] = s2[i]; }
#pragma unroll
      for (int u = 0; u < PF; ++u) { const int i = c.tid + u * BNS_THREADS; if (i < cnt) { d[i] = v1[u]; d[BNS_CHUNK * 64 + i] = v2[u]; } }
    }
    __syncthreads();
    BNS_T(c, 1);
  }
  int bias_off = 0;
  for (int l = 0; l < L; ++l) {
    const int T = n.T[l], MT = n.MT[l], M = n.K[l + 1];
    // Stage the fragments (mt0 .., t0 ..) of this layer: pair (mt, t) lands at ((mt - mt0) * nt + (t - t0)) * 64 + lane.
    // Chunks of at most BNS_CHUNK pairs keep the stage at 32 KB, so that two workgroups share a CU and one computes
    // while the other waits for its fragments.
    auto stage = [&](int mt0, int nmt, int t0, int nt) {
      if (whole) return;
      const f32x4 *s1 = (const f32x4 *)(lf + n.foff[l]), *s2 = (const f32x4 *)(dw + n.foff[l]);
      f32x4 *d = (f32x4 *)c.stage;
      const int cnt = nmt * nt * 64;
      BNS_T(c, (l == 0 ? 2 : (l == L - 1 ? 4 : 3)));
      __syncthreads();
      if (nt == T) {            // whole output tiles: one contiguous run of fragments; all loads in flight before the writes
        const f32x4 *r1 = s1 + mt0 * T * 64, *r2 = s2 + mt0 * T * 64;
        constexpr int PF = (BNS_CHUNK * 64 + BNS_THREADS - 1) / BNS_THREADS;
        f32x4 v1[PF], v2[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) { const int i = min(c.tid + u * BNS_THREADS, cnt - 1); v1[u] = r1[i]; v2[u] = r2[i]; }
#pragma unroll
        for (int u = 0; u < PF; ++u) { const int i = c.tid + u * BNS_THREADS; if (i < cnt) { d[i] = v1[u]; d[BNS_CHUNK * 64 + i] = v2[u]; } }
      } else {
        for (int i = c.tid; i < cnt; i += BNS_THREADS) {
          const int pair = i >> 6, mt = mt0 + pair / nt, t = t0 + pair % nt;
          const int src = (mt * T + t) * 64 + (i & 63);
          d[i] = s1[src]; d[BNS_CHUNK * 64 + i] = s2[src];
        }
      }
      __syncthreads();
      BNS_T(c, 1);
    };
    const f32x4 *LF = (const f32x4 *)c.stage + (whole ? (n.foff[l] >> 2) : 0), *DF = LF + BNS_CHUNK * 64;
    const f32x4 *BL = (const f32x4 *)(c.bn + 2 * BNS_MAXK + bias_off);   // bias of unit 16 mt + 4 g + r = BL[4 mt + g][r]
    bias_off += 16 * MT;
    if (l == 0) {
      f32x4 a1[BNS_R][BNS_MAXT], a2[BNS_R][BNS_MAXT];
#pragma unroll
      for (int rt = 0; rt < BNS_R; ++rt)
#pragma unroll
        for (int mt = 0; mt < BNS_MAXT; ++mt) { a1[rt][mt] = f32x4{0.f, 0.f, 0.f, 0.f}; a2[rt][mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      const int tc = whole ? T : BNS_CHUNK / BNS_MAXT;
      for (int t = 0; t < T; ++t) {
        const int t0 = t - t % tc, nt = min(tc, T - t0);
        if (t == t0) stage(0, MT, t0, nt);
        float hb[BNS_R][4], hs[BNS_R][4];
#pragma unroll
        for (int rt = 0; rt < BNS_R; ++rt) {
          const uint32_t w = bns_sgw(c, rt, n.sin_w[0] + (t >> 1));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int u = 16 * t + 4 * g + r;
            const float x = fmaf(T == 1 ? xin[rt][r] : in(rt, u), c.bn[u], c.bn[BNS_MAXK + u]);
            hb[rt][r] = x;
            hs[rt][r] = bns_flip(x, w, ((t & 1) << 4) + 4 * g + r);
          }
        }
#pragma unroll
        for (int mt = 0; mt < BNS_MAXT; ++mt)
          if (mt < MT) {
            const f32x4 fa = LF[(mt * nt + (t - t0)) * 64 + lane], fd = DF[(mt * nt + (t - t0)) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int rt = 0; rt < BNS_R; ++rt) {
                a1[rt][mt] = BGM_MFMA(fa[r], hb[rt][r], a1[rt][mt]);
                a2[rt][mt] = BGM_MFMA(fd[r], hs[rt][r], a2[rt][mt]);
              }
          }
      }
#pragma unroll
      for (int mt = 0; mt < BNS_MAXT; ++mt) {
        const f32x4 b = (mt < MT) ? BL[4 * mt + g] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < BNS_R; ++rt) {
          const uint32_t w = (mt < MT) ? bns_sgw(c, rt, n.sout_w[0] + (mt >> 1)) : 0u;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float y = a1[rt][mt][r] + b[r] + bns_flip(a2[rt][mt][r], w, ((mt & 1) << 4) + 4 * g + r);
            h[rt][mt][r] = (mt < MT) ? lrelu_s(y) : 0.0f;
          }
        }
      }
    } else {
      float hs[BNS_R][BNS_MAXT][4];
#pragma unroll
      for (int rt = 0; rt < BNS_R; ++rt)
#pragma unroll
        for (int t = 0; t < BNS_MAXT; ++t) {
          const uint32_t w = (t < T) ? bns_sgw(c, rt, n.sin_w[l] + (t >> 1)) : 0u;
#pragma unroll
          for (int r = 0; r < 4; ++r) hs[rt][t][r] = bns_flip(h[rt][t][r], w, ((t & 1) << 4) + 4 * g + r);
        }
      const bool last = (l == L - 1);
      float hn[BNS_R][BNS_MAXT][4];
      const int mc = BNS_CHUNK / BNS_MAXT;      // output tiles per staged chunk
      int mt0 = 0;
      auto tile = [&](int mt, f32x4 (&y)[BNS_R]) {
        f32x4 a1[BNS_R], a2[BNS_R];
#pragma unroll
        for (int rt = 0; rt < BNS_R; ++rt) { a1[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; a2[rt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int t = 0; t < BNS_MAXT; ++t)
          if (t < T) {
            const f32x4 fa = LF[((mt - mt0) * T + t) * 64 + lane], fd = DF[((mt - mt0) * T + t) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int rt = 0; rt < BNS_R; ++rt) {
                a1[rt] = BGM_MFMA(fa[r], h[rt][t][r], a1[rt]);
                a2[rt] = BGM_MFMA(fd[r], hs[rt][t][r], a2[rt]);
              }
          }
        const f32x4 b = BL[4 * mt + g];
#pragma unroll
        for (int rt = 0; rt < BNS_R; ++rt) {
          const uint32_t w = bns_sgw(c, rt, n.sout_w[l] + (mt >> 1));
#pragma unroll
          for (int r = 0; r < 4; ++r) y[rt][r] = a1[rt][r] + b[r] + bns_flip(a2[rt][r], w, ((mt & 1) << 4) + 4 * g + r);
        }
      };
      if (!last) {
        stage(0, MT, 0, T);
#pragma unroll
        for (int mt = 0; mt < BNS_MAXT; ++mt) {
          f32x4 y[BNS_R];
          if (mt < MT) tile(mt, y);
#pragma unroll
          for (int rt = 0; rt < BNS_R; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) hn[rt][mt][r] = (mt < MT) ? lrelu_s(y[rt][r]) : 0.0f;
        }
#pragma unroll
        for (int rt = 0; rt < BNS_R; ++rt)
#pragma unroll
          for (int t = 0; t < BNS_MAXT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[rt][t][r] = hn[rt][t][r];
      } else {
        pre(0);
        for (int mt = 0; mt < MT; ++mt) {
          if (mt % mc == 0) { mt0 = mt; stage(mt0, min(mc, MT - mt0), 0, T); }
          pre(mt + 1);          // operands of the NEXT tile's epilogue (e.g. the data row) are requested a tile ahead
          f32x4 y[BNS_R];
          tile(mt, y);
#pragma unroll
          for (int rt = 0; rt < BNS_R; ++rt) epi(rt, mt, y[rt]);
        }
      }
    }
    BNS_T(c, (l == 0 ? 2 : (l == L - 1 ? 4 : 3)));
  }
}

// ---------------------------------------------------------------------------------------------
// log-posterior of a block-structured panel / one Metropolis-Hastings iteration
// ---------------------------------------------------------------------------------------------
struct BnsMhArgs {
  float sig2[3];                       // fixed sigma_v^2, sigma_x^2, sigma_y^2 (params['sigma_*']); <= 0: the variance heads
  BnsNet net[4];                       // g, e, f, h (BNN_* ids); e unused here
  const float *theta, *lf;
  const float *dw;                     // [n_blocks * n_calls][set_floats]
  long long set_floats;
  const double *stats;                 // [n_blocks][2][2][64] of this iteration's parity (0: proposal, 1: current)
  const double *xstats;                // [n_blocks][2]
  const float *x, *y, *v;
  float *z;                            // current state [n x q] (updated in place by the MH mode)
  const float *zprop;                  // proposals [n x q]
  long long n, row_base;
  int q, p, z0, z1, z2, binary, bs, wg_per_block, block0;
  int n_items;                         // workgroups' worth of work: n_blocks * wg_per_block (the grid is rounded up to 8)
  int mode;                            // 0: log-posterior of z with the "current" statistics / call slot 0 -> out;  1: MH iteration
  int it;
  uint32_t k0, k1, stream0;            // mode 1: streams 2 it (proposal), 2 it + 1 (current); mode 0: stream0
  float *out;                          // mode 0: [n] log-posterior
  unsigned *acc_count;                 // mode 1 (optional): accepted proposals
  unsigned *acc_blocks;                // mode 1 (optional): accepted proposals of this iteration per block [n_blocks]
  unsigned long long *prof;            // -D BNS_PROF: cycles per phase, summed over workgroups (wave 0)
  // conditional latent prior (IdentifiableCausalBGM with batch statistics, bprior_api.hip): [n_states][n][q + 2] = mu [q], 1 / sigma^2,
  // (q / 2) log sigma^2 of every row for the state's own noisy call of the prior net (state 0: proposal / mode 0, state 1: current), or NULL = N(0, I)
  const float *prior;
  long long prior_stride;
};

__device__ __forceinline__ void bns_stat(const double *st, int col, double cnt, float &mean, float &var) {
  const double m = st[col] / cnt, v = st[64 + col] / cnt - m * m;
  mean = (float)m; var = (float)fmax(v, 0.0);
}

// -(NLL_v + NLL_x + NLL_y + |z|^2 / 2) of the rows of this wave for one state; zsrc = state, call slot `slot`.
__device__ __forceinline__ void bns_logpost_rows(const BnsCtx &c, const BnsMhArgs &a, int blk, int rib0, const long long (&row)[BNS_R],
                                                 const float *zsrc, int stat_slot, int dw_slot, int n_calls, uint32_t k1, uint32_t stream,
                                                 double cnt, float (&lp)[BNS_R]) {
  const int q = a.q, p = a.p, z0 = a.z0, z1 = a.z1, g = c.g;
  const double *st = a.stats + ((long long)blk * 2 + stat_slot) * 128;
  const double *xst = a.xstats + 2 * blk;
  const float *dwset = a.dw + ((long long)blk * n_calls + dw_slot) * a.set_floats;
  float xr[BNS_R], yr[BNS_R];
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) { xr[rt] = a.x[row[rt]]; yr[rt] = a.y[row[rt]]; }
  // ---- g: Gaussian likelihood of the covariates
  float ssq[BNS_R], raw[BNS_R], vcur[BNS_R][4], vnext[BNS_R][4];
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) {
    ssq[rt] = 0.0f; raw[rt] = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { vcur[rt][r] = 0.0f; vnext[rt][r] = 0.0f; }
  }
  bns_forward(c, a.net[BNN_G], a.theta, a.lf + a.net[BNN_G].fbase, dwset + a.net[BNN_G].dbase, a.k0, k1, stream, rib0,
              [&](int u, float &m, float &v) { bns_stat(st, u, cnt, m, v); },
              [&](int rt, int u) { return zsrc[row[rt] * q + min(u, q - 1)]; },     // columns >= q have zero scale and shift
              [&](int rt, int mt, const f32x4 &y) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int u = 16 * mt + 4 * g + r;
                  if (u < p) { const float d = vcur[rt][r] - y[r]; ssq[rt] = fmaf(d, d, ssq[rt]); }
                  else if (u == p) raw[rt] += y[r];
                }
              },
              [&](int mt) {      // data row one tile ahead: vcur <- vnext, vnext <- tile mt
#pragma unroll
                for (int rt = 0; rt < BNS_R; ++rt) {
                  const float *vr = a.v + row[rt] * p;
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    vcur[rt][r] = vnext[rt][r];
                    vnext[rt][r] = vr[min(16 * mt + 4 * g + r, p - 1)];
                  }
                }
              });
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) {
    const float s = sum_over_g(ssq[rt]), rw = sum_over_g(raw[rt]);
    const float s2 = a.sig2[0] > 0.0f ? a.sig2[0] : softplus_acc(rw) + BGM_EPS;
    lp[rt] = -(s / (2.0f * s2) + (float)p * logf(s2) * 0.5f);
  }
  // ---- h: treatment model, input (z0, z2)
  float mu[BNS_R];
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) { mu[rt] = 0.0f; raw[rt] = 0.0f; }
  bns_forward(c, a.net[BNN_H], a.theta, a.lf + a.net[BNN_H].fbase, dwset + a.net[BNN_H].dbase, a.k0, k1, stream, rib0,
              [&](int u, float &m, float &v) { bns_stat(st, u < z0 ? u : u + z1, cnt, m, v); },
              [&](int rt, int u) { return zsrc[row[rt] * q + min(u < z0 ? u : u + z1, q - 1)]; },
              [&](int rt, int mt, const f32x4 &y) { if (mt == 0 && g == 0) { mu[rt] += y[0]; raw[rt] += y[1]; } });
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) {
    const float m_ = sum_over_g(mu[rt]), rw = sum_over_g(raw[rt]);
    if (a.binary) lp[rt] -= fmaxf(m_, 0.0f) - m_ * xr[rt] + log1pf(expf(-fabsf(m_)));
    else { const float s2 = a.sig2[1] > 0.0f ? a.sig2[1] : softplus_acc(rw) + BGM_EPS, d = xr[rt] - m_; lp[rt] -= d * d / (2.0f * s2) + logf(s2) * 0.5f; }
  }
  // ---- f: outcome model, input (z0, z1, x)
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) { mu[rt] = 0.0f; raw[rt] = 0.0f; }
  bns_forward(c, a.net[BNN_F], a.theta, a.lf + a.net[BNN_F].fbase, dwset + a.net[BNN_F].dbase, a.k0, k1, stream, rib0,
              [&](int u, float &m, float &v) {
                if (u < z0 + z1) bns_stat(st, u, cnt, m, v);
                else { const double mm = xst[0] / cnt; m = (float)mm; v = (float)fmax(xst[1] / cnt - mm * mm, 0.0); }
              },
              [&](int rt, int u) { const float zv = zsrc[row[rt] * q + min(u, q - 1)]; return u < z0 + z1 ? zv : xr[rt]; },
              [&](int rt, int mt, const f32x4 &y) { if (mt == 0 && g == 0) { mu[rt] += y[0]; raw[rt] += y[1]; } });
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) {
    const float m_ = sum_over_g(mu[rt]), rw = sum_over_g(raw[rt]);
    const float s2 = a.sig2[2] > 0.0f ? a.sig2[2] : softplus_acc(rw) + BGM_EPS, d = yr[rt] - m_;
    lp[rt] -= d * d / (2.0f * s2) + logf(s2) * 0.5f;
    float zz = 0.0f;
    if (a.prior) {      // identifiable.py:541-551: -(|z - mu(u)|^2 / (2 sigma^2(u)) + (q / 2) log sigma^2(u))
      const float *pr = a.prior + (long long)dw_slot * a.prior_stride + row[rt] * (long long)(q + 2);
      for (int u = 0; u < q; ++u) { const float t = zsrc[row[rt] * q + u] - pr[u]; zz = fmaf(t, t, zz); }
      lp[rt] -= 0.5f * zz * pr[q] + pr[q + 1];
    } else {
      for (int u = 0; u < q; ++u) { const float t = zsrc[row[rt] * q + u]; zz = fmaf(t, t, zz); }
      lp[rt] -= 0.5f * zz;
    }
  }
}

static __global__ __launch_bounds__(BNS_THREADS, 4) void bns_mh_kernel(BnsMhArgs a) {
  extern __shared__ __attribute__((aligned(16))) float bns_lds[];
  BnsCtx c;
  c.tid = threadIdx.x; c.wave = c.tid >> 6; c.lane = c.tid & 63; c.j = c.lane & 15; c.g = c.lane >> 4;
  c.bn = bns_lds;
  c.sg = (uint32_t *)(bns_lds + 2 * BNS_MAXK + BNS_MAXB);
  c.stage = bns_lds + 2 * BNS_MAXK + BNS_MAXB + BNS_WAVES * BNS_R * 16 * BNS_SW;
  const int item = bns_xcd_item(a.n_items);
  if (item < 0) return;
  const int blk = item / a.wg_per_block, wib = item - blk * a.wg_per_block;
  const int rib0 = wib * BNS_ROWS;
  const long long blk_lo = (long long)blk * a.bs;
  const long long blk_n = min((long long)a.bs, a.n - blk_lo);
  long long row[BNS_R];
  bool valid[BNS_R];
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) {
    const long long r = rib0 + ((c.wave * BNS_R + rt) << 4) + c.j;
    valid[rt] = r < blk_n;
    row[rt] = blk_lo + (valid[rt] ? r : blk_n - 1);
  }
  const uint32_t k1 = a.k1 + (uint32_t)(a.block0 + blk);
  const double cnt = (double)blk_n;
#ifdef BNS_PROF
  c.t_last = clock64();
#endif
  if (a.mode == 0) {
    float lp[BNS_R];
    bns_logpost_rows(c, a, blk, rib0, row, a.z, 1, 0, 1, k1, a.stream0, cnt, lp);
#pragma unroll
    for (int rt = 0; rt < BNS_R; ++rt)
      if (valid[rt] && c.g == 0) a.out[row[rt]] = lp[rt];
    return;
  }
  float lpp[BNS_R], lpc[BNS_R];
  bns_logpost_rows(c, a, blk, rib0, row, a.zprop, 0, 0, 2, k1, 2u * (uint32_t)a.it, cnt, lpp);
  bns_logpost_rows(c, a, blk, rib0, row, a.z, 1, 1, 2, k1, 2u * (uint32_t)a.it + 1u, cnt, lpc);
  unsigned nacc = 0;
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) {
    const uint4 w4 = philox4x32_10((uint32_t)(a.row_base + row[rt]), (uint32_t)a.it >> 2, 0u, TAG_ACC, a.k0, a.k1);
    const int it = a.it;
    const unsigned w = (it & 2) ? ((it & 1) ? w4.w : w4.z) : ((it & 1) ? w4.y : w4.x);
    const bool acc = u01_open(w) < expf(fminf(lpp[rt] - lpc[rt], 0.0f));
    if (valid[rt] && acc) {
      for (int u = c.g; u < a.q; u += 4) a.z[row[rt] * a.q + u] = a.zprop[row[rt] * a.q + u];
      if (c.g == 0) ++nacc;
    }
  }
  if (a.acc_count || a.acc_blocks) {
    for (int off = 32; off > 0; off >>= 1) nacc += __shfl_xor(nacc, off);
    __shared__ unsigned acc_part[BNS_WAVES];
    if (c.lane == 0) acc_part[c.wave] = nacc;
    __syncthreads();
    if (c.tid == 0) {
      unsigned t = 0;
      for (int w = 0; w < BNS_WAVES; ++w) t += acc_part[w];
      if (t && a.acc_count) atomicAdd(a.acc_count, t);
      if (t && a.acc_blocks) atomicAdd(&a.acc_blocks[blk], t);
    }
  }
#ifdef BNS_PROF
  BNS_T(c, 5);
  if (c.tid == 0 && a.prof)
    for (int k = 0; k < 6; ++k) atomicAdd(&a.prof[k], c.tp[k]);
#endif
}

// ---------------------------------------------------------------------------------------------
// causal effects: f-net at counterfactual treatments on one state per row
//   infer_from_latent_posterior, base.py:671-763 (one kept draw per launch) and the dose loops of evaluate, :553-568.
// The treatment column is constant over the batch, so its batch-normalised value is 0 * gamma + beta whatever the dose
// (mean = dose, variance = 0): this is what the reference computes with use_bnn, and what is computed here.
// ---------------------------------------------------------------------------------------------
struct BnsEffArgs {
  float sig2_y;                        // fixed sigma_y^2 (params['sigma_y']); <= 0: the variance head
  BnsNet f;                            // dbase = 0: sets hold the outcome net only
  const float *theta, *lf;             // lf: packed loc of all nets (f.fbase applies)
  const float *dw;                     // [n_blocks * n_doses][set_floats]
  long long set_floats;
  const double *stats;                 // [n_blocks][2][2][64]; slot 1 = statistics of z
  const float *z;                      // [n x q]
  long long n, row_base;
  int q, z0, z1, bs, wg_per_block, block0, n_doses, n_items;
  const float *xvals;                  // [n_doses]
  uint32_t k0, k1, stream0;            // dose k uses noise stream stream0 + k
  int sample_y;                        // 1: y ~ N(mu, s2) with outcome noise (row, it_noise, k >> 2, TAG_YNOISE)[k & 3]
  uint32_t it_noise;
  double *sum_out; long long sum_stride;   // dose sums over rows: sum_out[k * sum_stride] += ...   (may be NULL)
  float *ite_out; long long ite_stride;    // n_doses == 2: ite_out[row * ite_stride] = y(x_0) - y(x_1)  (may be NULL)
  unsigned long long *prof;                // -D BNS_PROF
};
#define BNS_EFF_DOSES 256

// ---- pipelined dose loop of the effects kernel -------------------------------------------------
// Over the doses of one kept draw only the perturbation set and the sign words change: the input normalisation (the treatment
// column normalises to beta whatever the dose), the biases, the packed loc fragments and the rows' inputs are the same.  They
// are set up once per workgroup; per dose the next set's fragments (16 KB) are requested before the current dose is computed and
// stored behind it into the other half of a double buffer, the next dose's sign words are drawn into the other half of the sign
// buffer, and ONE barrier separates two doses (three, and a full prologue, in the generic routine).  Arithmetic and its order
// are those of bns_forward: identical results.
// Requirements (else the generic loop runs): the outcome net's input fits one k-tile, its fragments one chunk, its sign words
// half a row of the sign buffer.
__host__ __device__ inline bool bns_eff_fast_ok(const BnsNet &n) {
  return BNS_R == 1 && n.T[0] == 1 && n.n_layers >= 2 && n.foff[n.n_layers] <= BNS_CHUNK * 256 && n.swords <= BNS_SW / 2;
}
// one Flipout forward of the staged net for this wave's 16 rows: LF / DF = loc / perturbation fragments of the whole net in LDS,
// sgr = this lane's row of sign words, hb0 = the normalised input (k-tile 0).  epi(mt, y) receives the last layer's tiles.
template <class Epi>
__device__ __forceinline__ void bns_eff_net(const BnsCtx &c, const BnsNet &n, const f32x4 *LFn, const f32x4 *DFn, const uint32_t *sgr,
                                            const float (&hb0)[4], Epi epi) {
  const int L = n.n_layers, g = c.g, lane = c.lane;
  float h[BNS_MAXT][4];
  int bias_off = 0;
  for (int l = 0; l < L; ++l) {
    const int T = n.T[l], MT = n.MT[l];
    const f32x4 *LF = LFn + (n.foff[l] >> 2), *DF = DFn + (n.foff[l] >> 2);
    const f32x4 *BL = (const f32x4 *)(c.bn + 2 * BNS_MAXK + bias_off);
    bias_off += 16 * MT;
    if (l == 0) {
      f32x4 a1[BNS_MAXT], a2[BNS_MAXT];
#pragma unroll
      for (int mt = 0; mt < BNS_MAXT; ++mt) { a1[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; a2[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      float hs[4];
      {
        const uint32_t w = sgr[n.sin_w[0]];
#pragma unroll
        for (int r = 0; r < 4; ++r) hs[r] = bns_flip(hb0[r], w, 4 * g + r);
      }
#pragma unroll
      for (int mt = 0; mt < BNS_MAXT; ++mt)
        if (mt < MT) {
          const f32x4 fa = LF[mt * 64 + lane], fd = DF[mt * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            a1[mt] = BGM_MFMA(fa[r], hb0[r], a1[mt]);
            a2[mt] = BGM_MFMA(fd[r], hs[r], a2[mt]);
          }
        }
#pragma unroll
      for (int mt = 0; mt < BNS_MAXT; ++mt) {
        const f32x4 b = (mt < MT) ? BL[4 * mt + g] : f32x4{0.f, 0.f, 0.f, 0.f};
        const uint32_t w = (mt < MT) ? sgr[n.sout_w[0] + (mt >> 1)] : 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y = a1[mt][r] + b[r] + bns_flip(a2[mt][r], w, ((mt & 1) << 4) + 4 * g + r);
          h[mt][r] = (mt < MT) ? lrelu_s(y) : 0.0f;
        }
      }
    } else {
      float hs[BNS_MAXT][4];
#pragma unroll
      for (int t = 0; t < BNS_MAXT; ++t) {
        const uint32_t w = (t < T) ? sgr[n.sin_w[l] + (t >> 1)] : 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) hs[t][r] = bns_flip(h[t][r], w, ((t & 1) << 4) + 4 * g + r);
      }
      const bool last = (l == L - 1);
      auto tile = [&](int mt, f32x4 &y) {
        f32x4 a1 = f32x4{0.f, 0.f, 0.f, 0.f}, a2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < BNS_MAXT; ++t)
          if (t < T) {
            const f32x4 fa = LF[(mt * T + t) * 64 + lane], fd = DF[(mt * T + t) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              a1 = BGM_MFMA(fa[r], h[t][r], a1);
              a2 = BGM_MFMA(fd[r], hs[t][r], a2);
            }
          }
        const f32x4 b = BL[4 * mt + g];
        const uint32_t w = sgr[n.sout_w[l] + (mt >> 1)];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = a1[r] + b[r] + bns_flip(a2[r], w, ((mt & 1) << 4) + 4 * g + r);
      };
      if (!last) {
        float hn[BNS_MAXT][4];
#pragma unroll
        for (int mt = 0; mt < BNS_MAXT; ++mt) {
          f32x4 y;
          if (mt < MT) tile(mt, y);
#pragma unroll
          for (int r = 0; r < 4; ++r) hn[mt][r] = (mt < MT) ? lrelu_s(y[r]) : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < BNS_MAXT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) h[t][r] = hn[t][r];
      } else {
        for (int mt = 0; mt < MT; ++mt) {
          f32x4 y;
          tile(mt, y);
          epi(mt, y);
        }
      }
    }
  }
}
// The default outcome net [in <= 16] -> 64 -> 32 -> 8 -> [out <= 16] with every extent a compile-time constant: straight-line code, sign
// flips and epilogues only for the tiles that exist (the run-time-shaped routine above spends 6.5 VALU instructions per MFMA here)
__host__ __device__ inline bool bns_eff_default_ok(const BnsNet &n) {
  return bns_eff_fast_ok(n) && n.n_layers == 4 && n.MT[0] == 4 && n.MT[1] == 2 && n.MT[2] == 1 && n.MT[3] == 1;
}
template <int KT, int NT>
__device__ __forceinline__ void bns_eff_layer(const f32x4 *LF, const f32x4 *DF, const f32x4 *BL, const uint32_t *sgr, int sin_w, int sout_w, int lane,
                                              int g, const float (&h)[KT][4], float (&y)[NT][4]) {
  float hs[KT][4];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const uint32_t w = sgr[sin_w + (t >> 1)];
#pragma unroll
    for (int r = 0; r < 4; ++r) hs[t][r] = bns_flip(h[t][r], w, ((t & 1) << 4) + 4 * g + r);
  }
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
    f32x4 a1 = f32x4{0.f, 0.f, 0.f, 0.f}, a2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const f32x4 fa = LF[(mt * KT + t) * 64 + lane], fd = DF[(mt * KT + t) * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a1 = BGM_MFMA(fa[r], h[t][r], a1);
        a2 = BGM_MFMA(fd[r], hs[t][r], a2);
      }
    }
    const f32x4 b = BL[4 * mt + g];
    const uint32_t w = sgr[sout_w + (mt >> 1)];
#pragma unroll
    for (int r = 0; r < 4; ++r) y[mt][r] = a1[r] + b[r] + bns_flip(a2[r], w, ((mt & 1) << 4) + 4 * g + r);
  }
}
template <int NT>
__device__ __forceinline__ void bns_eff_act(float (&y)[NT][4]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) y[t][r] = lrelu_s(y[t][r]);
}
template <int T1, int T2, int T3, class Epi>
__device__ __forceinline__ void bns_eff_net4(const BnsCtx &c, const BnsNet &n, const f32x4 *LFn, const f32x4 *DFn, const uint32_t *sgr,
                                             const float (&hb0)[4], Epi epi) {
  const int g = c.g, lane = c.lane;
  const f32x4 *BL = (const f32x4 *)(c.bn + 2 * BNS_MAXK);
  float h0[1][4], h1[T1][4], h2[T2][4], h3[T3][4], y[1][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) h0[0][r] = hb0[r];
  bns_eff_layer<1, T1>(LFn + (n.foff[0] >> 2), DFn + (n.foff[0] >> 2), BL, sgr, n.sin_w[0], n.sout_w[0], lane, g, h0, h1);
  bns_eff_act<T1>(h1);
  bns_eff_layer<T1, T2>(LFn + (n.foff[1] >> 2), DFn + (n.foff[1] >> 2), BL + 4 * T1, sgr, n.sin_w[1], n.sout_w[1], lane, g, h1, h2);
  bns_eff_act<T2>(h2);
  bns_eff_layer<T2, T3>(LFn + (n.foff[2] >> 2), DFn + (n.foff[2] >> 2), BL + 4 * (T1 + T2), sgr, n.sin_w[2], n.sout_w[2], lane, g, h2, h3);
  bns_eff_act<T3>(h3);
  bns_eff_layer<T3, 1>(LFn + (n.foff[3] >> 2), DFn + (n.foff[3] >> 2), BL + 4 * (T1 + T2 + T3), sgr, n.sin_w[3], n.sout_w[3], lane, g, h3, y);
  epi(0, f32x4{y[0][0], y[0][1], y[0][2], y[0][3]});
}

// FAST 1: the pipelined dose loop (host: bns_eff_fast_ok(a.f)); 2: the same with the default outcome net's compile-time shape
// (bns_eff_default_ok); 0: one generic bns_forward per dose
template <int FAST>
static __global__ __launch_bounds__(BNS_THREADS, 4) void bns_effects_kernel(BnsEffArgs a) {
  extern __shared__ __attribute__((aligned(16))) float bns_lds[];
  __shared__ float dose_tot[BNS_WAVES][BNS_EFF_DOSES];
  BnsCtx c;
  c.tid = threadIdx.x; c.wave = c.tid >> 6; c.lane = c.tid & 63; c.j = c.lane & 15; c.g = c.lane >> 4;
  c.bn = bns_lds;
  c.sg = (uint32_t *)(bns_lds + 2 * BNS_MAXK + BNS_MAXB);
  c.stage = bns_lds + 2 * BNS_MAXK + BNS_MAXB + BNS_WAVES * BNS_R * 16 * BNS_SW;
  const int item = bns_xcd_item(a.n_items);
  if (item < 0) return;
  const int blk = item / a.wg_per_block, wib = item - blk * a.wg_per_block;
  const int rib0 = wib * BNS_ROWS;
  const long long blk_lo = (long long)blk * a.bs;
  const long long blk_n = min((long long)a.bs, a.n - blk_lo);
  long long row[BNS_R];
  bool valid[BNS_R];
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) {
    const long long r = rib0 + ((c.wave * BNS_R + rt) << 4) + c.j;
    valid[rt] = r < blk_n;
    row[rt] = blk_lo + (valid[rt] ? r : blk_n - 1);
  }
  const uint32_t k1 = a.k1 + (uint32_t)(a.block0 + blk);
  const double cnt = (double)blk_n;
  const double *st = a.stats + ((long long)blk * 2 + 1) * 128;
  const int q = a.q, zz = a.z0 + a.z1, g = c.g;
#ifdef BNS_PROF
  c.t_last = clock64();
#endif
  float y0[BNS_R];
  f32x4 nz[BNS_R];
  if constexpr (FAST != 0) {
    const BnsNet &n = a.f;
    const int L = n.n_layers, j = c.j, lane = c.lane;
    const float *gamma = a.theta + n.goff, *beta = gamma + n.K[0];
    // ---- once per workgroup: scale / shift of the latent columns, scale and beta of the treatment column, biases, loc fragments
    for (int u = c.tid; u < 16; u += BNS_THREADS) {
      float sc = 0.0f, sh = 0.0f;
      if (u < n.K[0]) {
        float mean = 0.0f, var = 1.0f;
        if (!n.bn_fixed) { if (u < zz) bns_stat(st, u, cnt, mean, var); else var = 0.0f; }
        sc = gamma[u] / sqrtf(var + BNN_BN_EPS);
        sh = u < zz ? beta[u] - mean * sc : beta[u];      // treatment column: the dose enters below (mean = dose)
      }
      c.bn[u] = sc; c.bn[BNS_MAXK + u] = sh;
    }
    {
      int bo = 0;
      for (int l = 0; l < L; ++l) {
        const float *bias = a.theta + n.woff[l] + 2 * n.K[l] * n.K[l + 1];
        const int M = n.K[l + 1];
        for (int o = c.tid; o < 16 * n.MT[l]; o += BNS_THREADS) c.bn[2 * BNS_MAXK + bo + o] = bias[min(o, M - 1)] * (o < M ? 1.0f : 0.0f);
        bo += 16 * n.MT[l];
      }
    }
    const int cnt4 = n.foff[L] >> 2;
    constexpr int PF = (BNS_CHUNK * 64 + BNS_THREADS - 1) / BNS_THREADS;
    f32x4 *locs = (f32x4 *)c.stage, *dwb0 = locs + BNS_CHUNK * 64, *dwb1 = dwb0 + BNS_CHUNK * 64;
    {
      const f32x4 *s1 = (const f32x4 *)(a.lf + n.fbase);
#pragma unroll
      for (int u = 0; u < PF; ++u) { const int i = c.tid + u * BNS_THREADS; if (i < cnt4) locs[i] = s1[i]; }
    }
    const int calls = n.swords >> 2;
    const uint32_t srow = (uint32_t)(rib0 + (c.wave << 4) + j);
    uint32_t *sgrow = c.sg + ((c.wave << 4) + j) * BNS_SW;
    auto signs = [&](int k, int b) {
      for (int cc = g; cc < calls; cc += 4) {
        const uint4 w = philox4x32_10(srow, (uint32_t)cc | ((uint32_t)n.net_id << 16), a.stream0 + (uint32_t)k, BNN_TAG_SIGN, a.k0, k1);
        uint32_t *dst = sgrow + b * (BNS_SW / 2) + 4 * cc;
        dst[0] = w.x; dst[1] = w.y; dst[2] = w.z; dst[3] = w.w;
      }
    };
    auto dwset = [&](int k) { return (const f32x4 *)(a.dw + ((long long)blk * a.n_doses + k) * a.set_floats); };
    float zraw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) zraw[r] = a.z[row[0] * q + min(4 * g + r, q - 1)];
    signs(0, 0);
    {
      const f32x4 *s2 = dwset(0);
#pragma unroll
      for (int u = 0; u < PF; ++u) { const int i = c.tid + u * BNS_THREADS; if (i < cnt4) dwb0[i] = s2[i]; }
    }
    __syncthreads();
    float hbz[4], scx = 0.0f, bex = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int u = 4 * g + r;
      hbz[r] = fmaf(zraw[r], c.bn[u], c.bn[BNS_MAXK + u]);
    }
    scx = c.bn[min(zz, 15)]; bex = c.bn[BNS_MAXK + min(zz, 15)];
    BNS_T(c, 0);
    for (int k = 0; k < a.n_doses; ++k) {
      const int b = k & 1;
      const bool more = k + 1 < a.n_doses;
      const float xv = a.xvals[k];
      f32x4 nx[PF];
      if (more) {
        const f32x4 *s2 = dwset(k + 1);
#pragma unroll
        for (int u = 0; u < PF; ++u) nx[u] = s2[min(c.tid + u * BNS_THREADS, cnt4 - 1)];
        signs(k + 1, b ^ 1);
      }
      if (a.sample_y && (k & 3) == 0)
        nz[0] = box_muller4(philox4x32_10((uint32_t)(a.row_base + row[0]), a.it_noise, (uint32_t)(k >> 2), TAG_YNOISE, a.k0, a.k1));
      BNS_T(c, 0);
      float hb0[4];
      const float shx = n.bn_fixed ? bex : bex - xv * scx;      // shift of the treatment column for this dose (batch statistics: mean = dose, variance = 0)
#pragma unroll
      for (int r = 0; r < 4; ++r) hb0[r] = (4 * g + r == zz) ? fmaf(xv, scx, shx) : hbz[r];
      float mu = 0.0f, raw = 0.0f;
      auto out = [&](int mt, const f32x4 &y) { if (mt == 0 && g == 0) { mu += y[0]; raw += y[1]; } };
      if constexpr (FAST == 2) bns_eff_net4<4, 2, 1>(c, n, locs, b ? dwb1 : dwb0, sgrow + b * (BNS_SW / 2), hb0, out);
      else bns_eff_net(c, n, locs, b ? dwb1 : dwb0, sgrow + b * (BNS_SW / 2), hb0, out);
      BNS_T(c, 3);
      float yk = sum_over_g(mu);
      if (a.sample_y) {
        const float s2 = a.sig2_y > 0.0f ? a.sig2_y : softplus_acc(sum_over_g(raw)) + BGM_EPS;
        const int e = k & 3;
        yk = fmaf(sqrtf(s2), e == 0 ? nz[0][0] : e == 1 ? nz[0][1] : e == 2 ? nz[0][2] : nz[0][3], yk);
      }
      float tot = (valid[0] && g == 0) ? yk : 0.0f;
      if (a.ite_out) {
        if (k == 0) y0[0] = yk;
        else if (k == 1 && valid[0] && g == 0) a.ite_out[row[0] * a.ite_stride] = y0[0] - yk;
      }
      if (a.sum_out) {
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
        if (lane == 0) {
          if (k < BNS_EFF_DOSES) dose_tot[c.wave][k] = tot;
          else atomicAdd(&a.sum_out[(long long)k * a.sum_stride], (double)tot);
        }
      }
      BNS_T(c, 5);
      if (more) {
        f32x4 *d = b ? dwb0 : dwb1;
#pragma unroll
        for (int u = 0; u < PF; ++u) { const int i = c.tid + u * BNS_THREADS; if (i < cnt4) d[i] = nx[u]; }
      }
      __syncthreads();
      BNS_T(c, 1);
    }
  } else
  for (int k = 0; k < a.n_doses; ++k) {
    const float xv = a.xvals[k];
    if (a.sample_y && (k & 3) == 0) {
#pragma unroll
      for (int rt = 0; rt < BNS_R; ++rt)
        nz[rt] = box_muller4(philox4x32_10((uint32_t)(a.row_base + row[rt]), a.it_noise, (uint32_t)(k >> 2), TAG_YNOISE, a.k0, a.k1));
    }
    float mu[BNS_R], raw[BNS_R];
#pragma unroll
    for (int rt = 0; rt < BNS_R; ++rt) { mu[rt] = 0.0f; raw[rt] = 0.0f; }
    bns_forward(c, a.f, a.theta, a.lf + a.f.fbase, a.dw + ((long long)blk * a.n_doses + k) * a.set_floats, a.k0, k1, a.stream0 + (uint32_t)k,
                rib0,
                [&](int u, float &m, float &v) { if (u < zz) bns_stat(st, u, cnt, m, v); else { m = xv; v = 0.0f; } },
                [&](int rt, int u) { const float zv = a.z[row[rt] * q + min(u, q - 1)]; return u < zz ? zv : xv; },
                [&](int rt, int mt, const f32x4 &y) { if (mt == 0 && g == 0) { mu[rt] += y[0]; raw[rt] += y[1]; } });
    float tot = 0.0f;
#pragma unroll
    for (int rt = 0; rt < BNS_R; ++rt) {
      float yk = sum_over_g(mu[rt]);
      if (a.sample_y) {
        const float s2 = a.sig2_y > 0.0f ? a.sig2_y : softplus_acc(sum_over_g(raw[rt])) + BGM_EPS;
        const int e = k & 3;
        yk = fmaf(sqrtf(s2), e == 0 ? nz[rt][0] : e == 1 ? nz[rt][1] : e == 2 ? nz[rt][2] : nz[rt][3], yk);
      }
      if (valid[rt] && g == 0) tot += yk;
      if (a.ite_out) {
        if (k == 0) y0[rt] = yk;
        else if (k == 1 && valid[rt] && g == 0) a.ite_out[row[rt] * a.ite_stride] = y0[rt] - yk;
      }
    }
    if (a.sum_out) {
      for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
      if (c.lane == 0) {
        if (k < BNS_EFF_DOSES) dose_tot[c.wave][k] = tot;
        else atomicAdd(&a.sum_out[(long long)k * a.sum_stride], (double)tot);
      }
    }
  }
#ifdef BNS_PROF
  BNS_T(c, 5);
  if (c.tid == 0 && a.prof)
    for (int k = 0; k < 6; ++k) atomicAdd(&a.prof[k], c.tp[k]);
#endif
  if (a.sum_out) {      // one atomic per workgroup and dose (63 000 atomics per address serialise in the L2 otherwise)
    __syncthreads();
    for (int k = c.tid; k < min(a.n_doses, BNS_EFF_DOSES); k += BNS_THREADS) {
      double t = 0.0;
      for (int w = 0; w < BNS_WAVES; ++w) t += (double)dose_tot[w][k];
      atomicAdd(&a.sum_out[(long long)k * a.sum_stride], t);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// evaluate (base.py:534-570): the whole panel is ONE batch
// ---------------------------------------------------------------------------------------------
// column sums / sums of squares of an [n x d] matrix into doubles st[0..d), st[BNS_MAXK..BNS_MAXK + d)
static __global__ __launch_bounds__(256) void bns_colstats_kernel(const float *m, long long n, int d, double *st) {
  __shared__ double red[4][2];
  for (int col = blockIdx.y; col < d; col += gridDim.y) {
    double s = 0.0, s2 = 0.0;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < n; r += (long long)gridDim.x * 256) {
      const double x = (double)m[r * d + col];
      s += x; s2 += x * x;
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); s2 += __shfl_xor(s2, off); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s; red[threadIdx.x >> 6][1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(&st[col], red[0][0] + red[1][0] + red[2][0] + red[3][0]);
      atomicAdd(&st[BNS_MAXK + col], red[0][1] + red[1][1] + red[2][1] + red[3][1]);
    }
  }
}

struct BnsEvalArgs {
  BnsNet net[4];
  const float *theta, *lf, *dw;        // one perturbation set [g | h | f | e]
  const double *stats;                 // statistics of z: [1 block][2][2][64], slot 1
  const double *xstats, *vstats;       // [2]; [2][BNS_MAXK]
  const float *x, *y, *v;
  float *z;                            // mode 0: written (e(v)); mode 1: read
  long long n;
  int q, p, z0, z1, z2, binary, mode;  // mode 0: z = e(v);  mode 1: reconstruction errors of g, h, f
  uint32_t k0, k1, stream;
  double *sums;                        // mode 1: [sum (v - v^)^2, sum (x - x^)^2, sum (y - y^)^2]
};
static __global__ __launch_bounds__(BNS_THREADS, 4) void bns_eval_kernel(BnsEvalArgs a) {
  extern __shared__ __attribute__((aligned(16))) float bns_lds[];
  BnsCtx c;
  c.tid = threadIdx.x; c.wave = c.tid >> 6; c.lane = c.tid & 63; c.j = c.lane & 15; c.g = c.lane >> 4;
  c.bn = bns_lds;
  c.sg = (uint32_t *)(bns_lds + 2 * BNS_MAXK + BNS_MAXB);
  c.stage = bns_lds + 2 * BNS_MAXK + BNS_MAXB + BNS_WAVES * BNS_R * 16 * BNS_SW;
  const int rib0 = blockIdx.x * BNS_ROWS;
  long long row[BNS_R];
  bool valid[BNS_R];
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) {
    const long long r = (long long)rib0 + ((c.wave * BNS_R + rt) << 4) + c.j;
    valid[rt] = r < a.n;
    row[rt] = valid[rt] ? r : a.n - 1;
  }
  const double cnt = (double)a.n;
  const int q = a.q, p = a.p, z0 = a.z0, z1 = a.z1, g = c.g;
  if (a.mode == 0) {
    bns_forward(c, a.net[BNN_E], a.theta, a.lf + a.net[BNN_E].fbase, a.dw + a.net[BNN_E].dbase, a.k0, a.k1, a.stream, rib0,
                [&](int u, float &m, float &v) {
                  const double mm = a.vstats[u] / cnt;
                  m = (float)mm; v = (float)fmax(a.vstats[BNS_MAXK + u] / cnt - mm * mm, 0.0);
                },
                [&](int rt, int u) { return u < p ? a.v[row[rt] * p + u] : 0.0f; },
                [&](int rt, int mt, const f32x4 &y) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    const int u = 16 * mt + 4 * g + r;
                    if (u < q && valid[rt]) a.z[row[rt] * q + u] = y[r];
                  }
                });
    return;
  }
  const double *st = a.stats + 128;
  float ev[BNS_R], mu[BNS_R];
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) { ev[rt] = 0.0f; mu[rt] = 0.0f; }
  bns_forward(c, a.net[BNN_G], a.theta, a.lf + a.net[BNN_G].fbase, a.dw + a.net[BNN_G].dbase, a.k0, a.k1, a.stream, rib0,
              [&](int u, float &m, float &v) { bns_stat(st, u, cnt, m, v); },
              [&](int rt, int u) { return u < q ? a.z[row[rt] * q + u] : 0.0f; },
              [&](int rt, int mt, const f32x4 &y) {
                const float *vr = a.v + row[rt] * p;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int u = 16 * mt + 4 * g + r;
                  if (u < p) { const float d = vr[u] - y[r]; ev[rt] = fmaf(d, d, ev[rt]); }
                }
              });
  float sv = 0.0f, sx = 0.0f, sy = 0.0f;
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) { const float t = sum_over_g(ev[rt]); if (valid[rt] && g == 0) sv += t; }
  bns_forward(c, a.net[BNN_H], a.theta, a.lf + a.net[BNN_H].fbase, a.dw + a.net[BNN_H].dbase, a.k0, a.k1, a.stream, rib0,
              [&](int u, float &m, float &v) { bns_stat(st, u < z0 ? u : u + z1, cnt, m, v); },
              [&](int rt, int u) { return u < a.net[BNN_H].K[0] ? a.z[row[rt] * q + (u < z0 ? u : u + z1)] : 0.0f; },
              [&](int rt, int mt, const f32x4 &y) { if (mt == 0 && g == 0) mu[rt] += y[0]; });
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) {
    float m_ = sum_over_g(mu[rt]);
    if (a.binary) m_ = sigmoid_f(m_);
    const float d = a.x[row[rt]] - m_;
    if (valid[rt] && g == 0) sx += d * d;
    mu[rt] = 0.0f;
  }
  bns_forward(c, a.net[BNN_F], a.theta, a.lf + a.net[BNN_F].fbase, a.dw + a.net[BNN_F].dbase, a.k0, a.k1, a.stream, rib0,
              [&](int u, float &m, float &v) {
                if (u < z0 + z1) bns_stat(st, u, cnt, m, v);
                else { const double mm = a.xstats[0] / cnt; m = (float)mm; v = (float)fmax(a.xstats[1] / cnt - mm * mm, 0.0); }
              },
              [&](int rt, int u) { return u < z0 + z1 ? a.z[row[rt] * q + u] : (u == z0 + z1 ? a.x[row[rt]] : 0.0f); },
              [&](int rt, int mt, const f32x4 &y) { if (mt == 0 && g == 0) mu[rt] += y[0]; });
#pragma unroll
  for (int rt = 0; rt < BNS_R; ++rt) {
    const float d = a.y[row[rt]] - sum_over_g(mu[rt]);
    if (valid[rt] && g == 0) sy += d * d;
  }
  for (int off = 32; off > 0; off >>= 1) { sv += __shfl_xor(sv, off); sx += __shfl_xor(sx, off); sy += __shfl_xor(sy, off); }
  __shared__ float part[BNS_WAVES][3];
  if (c.lane == 0) { part[c.wave][0] = sv; part[c.wave][1] = sx; part[c.wave][2] = sy; }
  __syncthreads();
  if (c.tid < 3) {
    double t = 0.0;
    for (int w = 0; w < BNS_WAVES; ++w) t += (double)part[w][c.tid];
    atomicAdd(&a.sums[c.tid], t);
  }
}
