// probe_bf16x3.hip -- measurement aid for the NEXT step of the MH kernel ("what comes next"): one hidden
// 64 -> 64 layer + LeakyReLU of the swapped-orientation MLP, chained `iters` times per wave, either
//   mode 0: fp32 MFMA (v_mfma_f32_16x16x4_f32), 64 MFMAs per layer and 16 chains -- what causal_mh_kernel does today, or
//   mode 1: split-precision bf16 x 3 (v_mfma_f32_16x16x32_bf16): W = W_hi + W_lo, h = h_hi + h_lo (bf16 each),
//           W h ~ W_hi h_hi + W_hi h_lo + W_lo h_hi: 24 MFMAs per layer; the accumulator -> B-operand chaining of the fp32
//           kernel carries over with a permuted K order (lane group g of K block T holds features 16 (2T + s) + 4 g + r).
// Reports time per layer and the result after `iters` layers so that the host can compare both modes with float64.
// Not on any product path.
#include <hip/hip_runtime.h>

#include "bgm_host.h"
#include "bgm_device.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
static __global__ __launch_bounds__(512) void bf16x3_probe_kernel(const float *W, const float *x0, float *out, int iters) {
  __shared__ __attribute__((aligned(16))) float wf[64 * 64];                 // fp32: A operand of step (t, r), tile mt at [((t*4+r)*4+mt)*64 + lane]
  __shared__ __attribute__((aligned(16))) bf16x8 whi[4 * 2 * 64], wlo[4 * 2 * 64];
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
  for (int e = tid; e < 64 * 64; e += 512) {
    const int l = e & 63, mt = (e >> 6) & 3, r = (e >> 8) & 3, t = e >> 10;
    wf[e] = W[(16 * mt + (l & 15)) * 64 + 16 * t + 4 * (l >> 4) + r];
  }
  for (int e = tid; e < 4 * 2 * 64; e += 512) {
    const int l = e & 63, T = (e >> 6) & 1, mt = e >> 7;
    bf16x8 hi, lo;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = u >> 2, r = u & 3;
      const float w = W[(16 * mt + (l & 15)) * 64 + 16 * (2 * T + s) + 4 * (l >> 4) + r];
      const __bf16 h_ = (__bf16)w;
      hi[u] = h_;
      lo[u] = (__bf16)(w - (float)h_);
    }
    whi[e] = hi; wlo[e] = lo;
  }
  __syncthreads();
  f32x4 h[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[t][r] = x0[i * 64 + 16 * t + 4 * g + r];      // chain i, feature 16 t + 4 g + r
  for (int it = 0; it < iters; ++it) {
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[mt] = BGM_MFMA(wf[((t * 4 + r) * 4 + mt) * 64 + lane], h[t][r], acc[mt]);
    } else {
#pragma unroll
      for (int T = 0; T < 2; ++T) {
        bf16x8 bh, bl;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float v = h[2 * T + (u >> 2)][u & 3];
          const __bf16 hh = (__bf16)v;
          bh[u] = hh;
          bl[u] = (__bf16)(v - (float)hh);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const bf16x8 ah = whi[(mt * 2 + T) * 64 + lane], al = wlo[(mt * 2 + T) * 64 + lane];
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[mt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[t][r] = lrelu(acc[t][r]);
  }
  if (blockIdx.x == 0 && tid < 64)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[i * 64 + 16 * t + 4 * g + r] = h[t][r];
}

/* mode 0: fp32 MFMA, 1: bf16 x 3.  W_host [64 x 64] (out-major), x_host [16 x 64]; out_host [16 x 64] = the activations after
 * `iters` layers; ns_per_layer = time of one layer of one wave's 16 chains with 8 waves on every CU. */
extern "C" int bgm_debug_bf16x3_probe(bgm_handle *h, int32_t mode, int32_t iters, const float *W_host, const float *x_host,
                                      float *out_host, double *ns_per_layer) {
  if (!h || iters <= 0 || !W_host || !x_host || !out_host) { bgm_set_error("bgm_debug_bf16x3_probe: bad argument"); return BGM_E_INVALID; }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  float *W, *x, *out;
  BGM_HIP_CHECK(hipMalloc(&W, 64 * 64 * 4)); BGM_HIP_CHECK(hipMalloc(&x, 16 * 64 * 4)); BGM_HIP_CHECK(hipMalloc(&out, 16 * 64 * 4));
  BGM_HIP_CHECK(hipMemcpy(W, W_host, 64 * 64 * 4, hipMemcpyHostToDevice));
  BGM_HIP_CHECK(hipMemcpy(x, x_host, 16 * 64 * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  BGM_HIP_CHECK(hipEventCreate(&e0)); BGM_HIP_CHECK(hipEventCreate(&e1));
  auto launch = [&](int n) {
    if (mode == 0) hipLaunchKernelGGL(bf16x3_probe_kernel<0>, dim3(h->n_cus), dim3(512), 0, 0, W, x, out, n);
    else hipLaunchKernelGGL(bf16x3_probe_kernel<1>, dim3(h->n_cus), dim3(512), 0, 0, W, x, out, n);
  };
  launch(iters);
  BGM_HIP_CHECK(hipEventRecord(e0, 0));
  launch(iters);
  BGM_HIP_CHECK(hipEventRecord(e1, 0));
  BGM_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  BGM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  BGM_HIP_CHECK(hipMemcpy(out_host, out, 16 * 64 * 4, hipMemcpyDeviceToHost));
  if (ns_per_layer) *ns_per_layer = (double)ms * 1e6 / (double)iters;
  hipFree(W); hipFree(x); hipFree(out); hipEventDestroy(e0); hipEventDestroy(e1);
  return BGM_OK;
}
