// fit_types.h -- plain argument structs of the fit kernels (shared with the host handle).
#pragma once
#include "causal_kernels.h"

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

struct DwLayer { long long a_off, d_off; int K, N, out_off; };  // widths (multiples of 16); offset in the partial
#define BGM_MAX_DW_LAYERS 32
struct DwArgs {
  const float *ws;
  float *partial;          // [n_slices][partial_stride]
  long long partial_stride;
  int B, rows_per_slice, n_layers;
  DwLayer layer[BGM_MAX_DW_LAYERS];
};
