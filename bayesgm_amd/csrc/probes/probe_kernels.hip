// probe_kernels.hip -- measurement aids (NOT part of libbgm_hip.so / include/bgm_hip.h): micro-benchmarks quoted in DESIGN_HISTORY.md section 4.
// Built into bayesgm_amd/csrc/probes/libbgm_probe.so by `python -m bayesgm_amd.csrc.build --probes`; used by scripts/probe_*.py only.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../bgm_device.h"

#define PROBE_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "probe: %s -> %s\n", #x, hipGetErrorString(e_)); return -1; } } while (0)

static int probe_n_cus(int device) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 256;
  return prop.multiProcessorCount;
}

// ---------------------------------------------------------------------------
// Measurement aid (bench / DESIGN.md): effective shader clock under an fp32-MFMA load.
// Every wave issues `iters` x 16 back-to-back v_mfma_f32_16x16x4_f32 on 4 independent accumulators;
// wave 0 of block 0 reports shader cycles (s_memtime) and the 100 MHz real-time counter.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512) void clock_probe_kernel(int iters, unsigned long long *out, float *sink) {
  f32x4 acc[4];
  for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = BGM_MFMA(a, b, acc[k]);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int k = 0; k < 4; ++k) s += acc[k][0];
  if (s == 123.456f) sink[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

// Micro-benchmark of the scheduled tile-group block alone: every wave loops over dense_group4_k64_asm (64 MFMAs +
// 16 ds_read_b128 from a 16 KiB LDS region) -- mode 1 -- or over 64 MFMAs with register operands only -- mode 0.
template <int MODE>
__global__ __launch_bounds__(512) void group_probe_kernel(int iters, float *sink) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 4 * 4096; i += blockDim.x) lds[i] = 1e-3f * (float)(i & 255);
  __syncthreads();
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, lane_off = 64 * g + j;
  f32x4 in[4], acc[4];
  for (int k = 0; k < 4; ++k) { in[k] = f32x4{1.f + lane * 1e-3f, 0.5f, 0.25f, 0.125f}; acc[k] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int i = 0; i < iters; ++i) {
    if constexpr (MODE == 1) {
      dense_group4_k64_asm(lds_byte_addr(lds + (i & 3) * 4096 + lane_off * 4), in, acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = BGM_MFMA(in[s >> 2][s & 3], in[k][s & 3], acc[k]);
    }
  }
  float t = 0.f;
  for (int k = 0; k < 4; ++k) t += acc[k][0];
  if (t == 123.456f) sink[0] = t;
}
extern "C" int bgm_probe_group(int device, int32_t mode, int32_t waves_per_cu, int32_t iters, double *mfma_tflops) {
  if (iters <= 0) return -2;
  const int n_cus = probe_n_cus(device);
  PROBE_CHECK(hipSetDevice(device));
  float *sink;
  PROBE_CHECK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1;
  PROBE_CHECK(hipEventCreate(&e0)); PROBE_CHECK(hipEventCreate(&e1));
  const int threads = 64 * waves_per_cu, ldsb = 4 * 4096 * 4;
  for (int rep = 0; rep < 2; ++rep) {
    if (rep == 1) PROBE_CHECK(hipEventRecord(e0, 0));
    if (mode == 1) hipLaunchKernelGGL(group_probe_kernel<1>, dim3(n_cus), dim3(threads), ldsb, 0, iters, sink);
    else hipLaunchKernelGGL(group_probe_kernel<0>, dim3(n_cus), dim3(threads), ldsb, 0, iters, sink);
  }
  PROBE_CHECK(hipEventRecord(e1, 0));
  PROBE_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  PROBE_CHECK(hipEventElapsedTime(&ms, e0, e1));
  if (mfma_tflops) *mfma_tflops = (double)n_cus * waves_per_cu * iters * 64.0 * 2048.0 / (ms * 1e-3) / 1e12;
  hipFree(sink); hipEventDestroy(e0); hipEventDestroy(e1);
  return 0;
}

extern "C" int bgm_probe_clock(int device, int32_t iters, double *shader_mhz, double *mfma_tflops) {
  if (iters <= 0) return -2;
  const int n_cus = probe_n_cus(device);
  PROBE_CHECK(hipSetDevice(device));
  unsigned long long *out; float *sink;
  PROBE_CHECK(hipMalloc(&out, 16)); PROBE_CHECK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1;
  PROBE_CHECK(hipEventCreate(&e0)); PROBE_CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(clock_probe_kernel, dim3(n_cus), dim3(512), 0, 0, iters / 8 + 1, out, sink);   // warm
  PROBE_CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(clock_probe_kernel, dim3(n_cus), dim3(512), 0, 0, iters, out, sink);
  PROBE_CHECK(hipEventRecord(e1, 0));
  PROBE_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  PROBE_CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long host[2];
  PROBE_CHECK(hipMemcpy(host, out, 16, hipMemcpyDeviceToHost));
  if (shader_mhz) *shader_mhz = (double)host[0] / ((double)host[1] / 100.0);   // real-time counter = 100 MHz
  if (mfma_tflops) *mfma_tflops = (double)n_cus * 8.0 * iters * 16.0 * 2048.0 / (ms * 1e-3) / 1e12;
  hipFree(out); hipFree(sink); hipEventDestroy(e0); hipEventDestroy(e1);
  return 0;
}

// probe_bf16x3.hip -- measurement aid for the NEXT step of the MH kernel (DESIGN_HISTORY.md section 4, "what comes next"): one hidden
// 64 -> 64 layer + LeakyReLU of the swapped-orientation MLP, chained `iters` times per wave, either
//   mode 0: fp32 MFMA (v_mfma_f32_16x16x4_f32), 64 MFMAs per layer and 16 chains -- what causal_mh_kernel does today, or
//   mode 1: split-precision bf16 x 3 (v_mfma_f32_16x16x32_bf16): W = W_hi + W_lo, h = h_hi + h_lo (bf16 each),
//           W h ~ W_hi h_hi + W_hi h_lo + W_lo h_hi: 24 MFMAs per layer; the accumulator -> B-operand chaining of the fp32
//           kernel carries over with a permuted K order (lane group g of K block T holds features 16 (2T + s) + 4 g + r).
// Reports time per layer and the result after `iters` layers so that the host can compare both modes with float64.
// Not on any product path.
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
extern "C" int bgm_probe_bf16x3(int device, int32_t mode, int32_t iters, const float *W_host, const float *x_host,
                                      float *out_host, double *ns_per_layer) {
  if (iters <= 0 || !W_host || !x_host || !out_host) return -2;
  const int n_cus = probe_n_cus(device);
  PROBE_CHECK(hipSetDevice(device));
  float *W, *x, *out;
  PROBE_CHECK(hipMalloc(&W, 64 * 64 * 4)); PROBE_CHECK(hipMalloc(&x, 16 * 64 * 4)); PROBE_CHECK(hipMalloc(&out, 16 * 64 * 4));
  PROBE_CHECK(hipMemcpy(W, W_host, 64 * 64 * 4, hipMemcpyHostToDevice));
  PROBE_CHECK(hipMemcpy(x, x_host, 16 * 64 * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  PROBE_CHECK(hipEventCreate(&e0)); PROBE_CHECK(hipEventCreate(&e1));
  auto launch = [&](int n) {
    if (mode == 0) hipLaunchKernelGGL(bf16x3_probe_kernel<0>, dim3(n_cus), dim3(512), 0, 0, W, x, out, n);
    else hipLaunchKernelGGL(bf16x3_probe_kernel<1>, dim3(n_cus), dim3(512), 0, 0, W, x, out, n);
  };
  launch(iters);
  PROBE_CHECK(hipEventRecord(e0, 0));
  launch(iters);
  PROBE_CHECK(hipEventRecord(e1, 0));
  PROBE_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  PROBE_CHECK(hipEventElapsedTime(&ms, e0, e1));
  PROBE_CHECK(hipMemcpy(out_host, out, 16 * 64 * 4, hipMemcpyDeviceToHost));
  if (ns_per_layer) *ns_per_layer = (double)ms * 1e6 / (double)iters;
  hipFree(W); hipFree(x); hipFree(out); hipEventDestroy(e0); hipEventDestroy(e1);
  return 0;
}
