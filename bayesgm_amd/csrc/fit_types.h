// fit_types.h -- plain argument structs of the fit kernels (shared with the host handle).
#pragma once
#include "causal_kernels.h"
#include "fit_sync.h"

// Workspace: offsets in floats of the per-batch-row buffers, each [B][width].
struct FitWs {
  int B;             // rows allocated
  long long zin;     // [B][16*KT1]   extended first-layer input (z, x, 0..) in natural feature order
  long long ag;      // [n_g_hidden][B][64]   h1..h5
  long long outg;    // [B][16*NTL]   g output (mu, s_raw)
  long long af1, af2, af3, outf;   // [B][64] [B][32] [B][16] [B][16]
  long long ah1, ah2, ah3, outh;
  long long dg;      // [n_g_hidden][B][64]   dpre of g hidden layers
  long long dgl;     // [B][16*NTL]
  long long df1, df2, df3, df4;    // [B][64] [B][32] [B][16] [B][16]
  long long dh1, dh2, dh3, dh4;
  long long dz;      // [B][q]
  long long total;
};

struct FitMeta {     // offsets (floats) into the TRANSPOSED blob used by the backward kernel
  int w1g, w1f, w1h; // [64][16*KT1]
  int wg;            // n_gh consecutive [64][64]
  int wgl;           // [16*NTL][64]
  int wf2, wf3, wf4; // [32][64], [16][32], [16][16]
  int wh2, wh3, wh4;
  int total;
};

struct FitKArgs {
  const float *blob;       // forward (fit_fwd_kernel) or transposed (fit_bwd_kernel) packed weights
  CausalMeta m;            // forward blob layout + model constants
  FitMeta bm;              // transposed blob layout
  FitWs ws;
  float *wsp;              // workspace base
  const float *x, *y, *v;  // full panels [N], [N], [N x p]
  const float *data_z;     // [N x q]
  const int *idx;          // [B] rows of this minibatch (NULL: rows row_lo + b)
  long long row_lo;
  int B;                   // rows in this (local) minibatch
  float inv_B;             // 1 / global batch size (batch-mean losses)
  int z_mode;              // 0: theta phase (store dpre), 1: z phase (store dz)
  double *loss;            // [8] accumulators: loss_v*B, sse_v, loss_x*B, sse_x|bce, loss_y*B, sse_y, loss_z*B
};

// Keras Adam (optimizer_v2) on canonical parameter c + scatter into the packed blobs / the transposed mirror.  One expression
// for the stand-alone kernel (fit_adam_theta_kernel) and the gradient-tile kernel's fused epilogue (fit_chain_dw_kernel).
struct FitAdamTheta {
  int on;                  // fit_chain_dw_kernel: apply the step on the tile's own parameters instead of storing the gradient
  float lr_t, b1, b2, eps;
  float *m1, *m2, *theta_out;        // theta_out: the buffer the new parameters go to (may be the one read)
  float *fwd_blob, *bwd_blob, *mirror;
  const int *fwd_dst, *fwd_dst2, *bwd_dst, *mirror_dst;
};
struct FitAdamPre { float th, m1, m2; int d0, d1, d2, d3; };      // parameter c's state, requested ahead of the gradient
__device__ __forceinline__ FitAdamPre fit_adam_theta_load(int c, const float *theta, const FitAdamTheta &ad) {
  FitAdamPre p;
  p.th = theta[c]; p.m1 = ad.m1[c]; p.m2 = ad.m2[c];
  p.d0 = ad.fwd_dst[c]; p.d1 = ad.fwd_dst2[c]; p.d2 = ad.bwd_dst[c]; p.d3 = ad.mirror ? ad.mirror_dst[c] : -1;
  return p;
}
__device__ __forceinline__ void fit_adam_theta_apply(int c, float g, const FitAdamPre &p, const FitAdamTheta &ad) {
  // every rounding spelled out: the stand-alone kernel and the fused epilogue must agree bit for bit whatever the compiler would contract
  const float m = __fmaf_rn(ad.b1, p.m1, __fmul_rn(1.0f - ad.b1, g));
  const float v = __fmaf_rn(ad.b2, p.m2, __fmul_rn(__fmul_rn(1.0f - ad.b2, g), g));
  ad.m1[c] = m;
  ad.m2[c] = v;
  const float w = __fsub_rn(p.th, __fdiv_rn(__fmul_rn(ad.lr_t, m), __fadd_rn(__fsqrt_rn(v), ad.eps)));
  ad.theta_out[c] = w;
  if (p.d0 >= 0) ad.fwd_blob[p.d0] = w;
  if (p.d1 >= 0) ad.fwd_blob[p.d1] = w;
  if (p.d2 >= 0) ad.bwd_blob[p.d2] = w;
  if (p.d3 >= 0) ad.mirror[p.d3] = w;      // transposed weights of the row-tile-chain kernels (fit_chain.h)
}
__device__ __forceinline__ void fit_adam_theta_one(int c, float g, const float *theta, const FitAdamTheta &ad) {
  fit_adam_theta_apply(c, g, fit_adam_theta_load(c, theta, ad), ad);
}

struct DwLayer { long long a_off, d_off; int K, N, out_off; };  // widths (multiples of 16); offset in the partial
#define BGM_MAX_DW_LAYERS 32
struct DwArgs {
  const float *ws;
  float *partial;          // [n_slices][partial_stride]
  long long partial_stride;
  int B, rows_per_slice, n_layers;
  DwLayer layer[BGM_MAX_DW_LAYERS];
};
