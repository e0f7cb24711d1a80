// aux_kernels.hip -- encoder forward, ADRF slot reduction, per-row mean/quantiles.
//
// replaces (src/bayesgm/models/causalbgm/base.py):
//   self.e_net(data_v)                       :479, :538   -> causal_encode_kernel
//   adrf_draw_sums / n_seen                  :660-663     -> adrf_reduce_kernel
//   np.mean / np.quantile(axis)              :640-642, :664-666 -> row_mean_quantiles_kernel
#include <algorithm>
#include <cmath>
#include <cstring>

#include "bgm_host.h"
#include "gx_host.h"

// ---------------------------------------------------------------------------
// Encoder  z = e(v):  p -> 64 x n_hidden -> q.  One wave = 16 rows; V rows are
// streamed once from HBM straight into the B-operand layout (feature 16t+4g+r).
// ---------------------------------------------------------------------------
struct EncMeta {
  int p, q, n_hh;         // n_hh hidden->hidden 64x64 layers
  int w1, b1, wh, bh, wl, bl, total;
};

template <int KTV, int NTQ, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void causal_encode_kernel(const float *blob, EncMeta m,
                                                                   const float *v, long long n, float *z) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  lds_fill(lds, blob, m.total);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4, lane_off = 64 * g + j;
  const long long n_tiles = (n + 15) / 16;
  for (long long tile = (long long)blockIdx.x * WAVES + wave; tile < n_tiles;
       tile += (long long)gridDim.x * WAVES) {
    BGM_NO_HOIST();
    const long long row0 = tile * 16;
    f32x4 vin[1][KTV];
    load_v_rows<KTV, 1>(v, n, m.p, row0, j, g, vin);
    f32x4 h[1][4];
    dense<KTV, 4, 4, 1>(lds + m.w1, lds + m.b1, lane_off, g, vin, h);
    lrelu_inplace<4, 1>(h);
    for (int l = 0; l < m.n_hh; ++l) {
      BGM_NO_HOIST();
      f32x4 h2[1][4];
      dense<4, 4, 4, 1>(lds + m.wh + l * 4096, lds + m.bh + l * 64, lane_off, g, h, h2);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[0][t][r] = lrelu(h2[0][t][r]);
    }
    f32x4 o[1][NTQ];
    dense<4, 4, NTQ, 1>(lds + m.wl, lds + m.bl, lane_off, g, h, o);
    const long long row = row0 + j;
    if (row < n) {
#pragma unroll
      for (int t = 0; t < NTQ; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * t + 4 * g + r;
          if (f < m.q) z[row * (long long)m.q + f] = o[0][t][r];
        }
    }
  }
}

static int build_eblob(bgm_handle *h, EncMeta &m, hipStream_t stream) {
  const HostNet &E = h->nets[BGM_NET_E];
  if (!E.set) { bgm_set_error("encoder weights not set"); return BGM_E_STATE; }
  const int p = h->p, q = h->q;
  const int KTV = bgm_enc_in_tiles(p), NTQ = (q + 15) / 16;
  if (KTV < 0 || NTQ > 2) { bgm_set_error("encoder: v_dim > 208 or sum(z_dims) > 32 not compiled"); return BGM_E_UNSUPPORTED; }
  std::memset(&m, 0, sizeof(m));
  m.p = p; m.q = q; m.n_hh = h->cfg.n_hidden_e - 1;
  int off = 0;
  auto take = [&](int n) { int o = off; off += (n + 3) / 4 * 4; return o; };
  m.w1 = take(16 * KTV * 64); m.b1 = take(64);
  m.wh = take(m.n_hh * 4096); m.bh = take(m.n_hh * 64);
  m.wl = take(64 * 16 * NTQ); m.bl = take(16 * NTQ);
  m.total = off;
  if ((size_t)m.total * 4 > 160 * 1024) { bgm_set_error("encoder does not fit the LDS-resident layout"); return BGM_E_UNSUPPORTED; }
  if (h->eblob_valid) return BGM_OK;
  std::vector<float> blob(m.total, 0.0f);
  auto ident = [](int rho) { return rho; };
  pack_layer(blob, m.w1, E.W(0), p, 64, KTV, 4, ident); pack_bias(blob, m.b1, E.b(0), 64, 4);
  for (int l = 0; l < m.n_hh; ++l) {
    pack_layer(blob, m.wh + l * 4096, E.W(1 + l), 64, 64, 4, 4, ident);
    pack_bias(blob, m.bh + l * 64, E.b(1 + l), 64, 4);
  }
  const int LE = (int)E.dims.size() - 2;
  pack_layer(blob, m.wl, E.W(LE), 64, q, 4, NTQ, ident); pack_bias(blob, m.bl, E.b(LE), q, NTQ);
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (h->eblob_cap < blob.size()) {
    if (h->eblob_dev) BGM_HIP_CHECK(hipFree(h->eblob_dev));
    BGM_HIP_CHECK(hipMalloc(&h->eblob_dev, blob.size() * sizeof(float)));
    h->eblob_cap = blob.size();
  }
  BGM_HIP_CHECK(hipMemcpyAsync(h->eblob_dev, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice, stream));
  BGM_HIP_CHECK(hipStreamSynchronize(stream));
  h->eblob_valid = true;
  return BGM_OK;
}

#define BGM_ENC_VARIANTS(X) X(13, 1) X(13, 2) X(7, 1) X(7, 2) X(2, 1) X(2, 2)
static constexpr int ENC_WAVES = 8;

extern "C" int bgm_causal_encode(bgm_handle *h, const float *v, int64_t n, float *z, void *stream_) {
  if (!h || !h->configured) { bgm_set_error("bgm_causal_encode: handle not configured"); return BGM_E_STATE; }
  if (n <= 0) return BGM_OK;
  if (!v || !z) { bgm_set_error("bgm_causal_encode: NULL pointer"); return BGM_E_INVALID; }
  hipStream_t stream = (hipStream_t)stream_;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (gx_enc_wanted(h)) return gx_encode(h, v, n, z, stream);     // e_units other than [64]*k, v_dim > 208: the general-width engine
  EncMeta m;
  int rc = build_eblob(h, m, stream);
  if (rc) return rc;
  const int KTV = bgm_enc_in_tiles(h->p), NTQ = (h->q + 15) / 16;
  const long long tiles = (n + 15) / 16;
  const int grid = (int)std::max<long long>(1, std::min<long long>((tiles + ENC_WAVES - 1) / ENC_WAVES, h->n_cus));
  const int lds = m.total * 4;
#define X(KTV_, NTQ_)                                                                             \
  if (KTV == KTV_ && NTQ == NTQ_) {                                                               \
    auto k = causal_encode_kernel<KTV_, NTQ_, ENC_WAVES>;                                         \
    BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * ENC_WAVES), lds, stream, h->eblob_dev, m, v, (long long)n, z); \
    BGM_HIP_CHECK(hipGetLastError());                                                             \
    return BGM_OK;                                                                                \
  }
  BGM_ENC_VARIANTS(X)
#undef X
  bgm_set_error("no compiled encoder variant for (KTV,NTQ)=(" + std::to_string(KTV) + "," + std::to_string(NTQ) + ")");
  return BGM_E_UNSUPPORTED;
}

// ---------------------------------------------------------------------------
// ADRF slot reduction (fixed summation order -> deterministic)
// ---------------------------------------------------------------------------
// partial [n_slots][n_keep][n_doses] (draw-major, as the sampling kernels accumulate it) -> out [n_doses][n_keep] (the reference's
// orientation of the dose-response draws, base.py:660)
__global__ void adrf_reduce_kernel(const float *partial, int n_slots, long long kd, int n_doses, int n_keep, double inv_n, float *out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kd) return;
  double s = 0.0;
  for (int sl = 0; sl < n_slots; ++sl) s += (double)partial[(long long)sl * kd + i];
  const long long d = i / n_doses;
  const int k = (int)(i - d * n_doses);
  out[(long long)k * n_keep + d] = (float)(s * inv_n);
}

extern "C" int bgm_adrf_reduce(bgm_handle *h, const float *partial, int32_t n_slots, int32_t n_doses,
                               int32_t n_keep, double n_total, float *out, void *stream_) {
  if (!h || !partial || !out || n_slots <= 0 || n_doses <= 0 || n_keep <= 0 || !(n_total > 0)) {
    bgm_set_error("bgm_adrf_reduce: bad argument"); return BGM_E_INVALID;
  }
  BGM_HIP_CHECK(hipSetDevice(h->device));
  const long long kd = (long long)n_doses * n_keep;
  hipLaunchKernelGGL(adrf_reduce_kernel, dim3((unsigned)((kd + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                     partial, n_slots, kd, n_doses, n_keep, 1.0 / n_total, out);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}

// ---------------------------------------------------------------------------
// per-row mean + two linear-interpolated quantiles (np.quantile default method):
// one 256-thread block per row, bitonic sort of the row in LDS.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float np_lerp(float a, float b, double t) {  // numpy _lerp
  const double d = (double)b - (double)a;
  const double r = (t >= 0.5) ? (double)b - d * (1.0 - t) : (double)a + d * t;
  return (float)r;
}

__global__ __launch_bounds__(256) void row_mean_quantiles_kernel(const float *in, long long n_rows, int m, int m_pow2,
                                                                double q_lo, double q_hi, float *mean,
                                                                float *lo, float *hi) {
  extern __shared__ __attribute__((aligned(16))) float srt[];
  __shared__ double red[256];
  for (long long row = blockIdx.x; row < n_rows; row += gridDim.x) {
    const float *src = in + row * (long long)m;
    double s = 0.0;
    for (int i = threadIdx.x; i < m_pow2; i += 256) {
      const float x = i < m ? src[i] : INFINITY;
      srt[i] = x;
      if (i < m) s += (double)x;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
      __syncthreads();
    }
    for (int k = 2; k <= m_pow2; k <<= 1) {
      for (int jj = k >> 1; jj > 0; jj >>= 1) {
        for (int i = threadIdx.x; i < m_pow2; i += 256) {
          const int ixj = i ^ jj;
          if (ixj > i) {
            const float a = srt[i], b = srt[ixj];
            const bool up = (i & k) == 0;
            if ((a > b) == up) { srt[i] = b; srt[ixj] = a; }
          }
        }
        __syncthreads();
      }
    }
    if (threadIdx.x == 0) {
      mean[row] = (float)(red[0] / (double)m);
      const double qs[2] = {q_lo, q_hi};
      float res[2];
      for (int t = 0; t < 2; ++t) {
        const double vi = qs[t] * (double)(m - 1);
        int i0 = (int)floor(vi);
        i0 = i0 < 0 ? 0 : (i0 > m - 1 ? m - 1 : i0);
        const int i1 = i0 + 1 > m - 1 ? m - 1 : i0 + 1;
        res[t] = np_lerp(srt[i0], srt[i1], vi - (double)i0);
      }
      lo[row] = res[0];
      hi[row] = res[1];
    }
    __syncthreads();
  }
}

// More than 32768 values per row: no sort, the order statistics the two quantiles need are SELECTED -- four 8-bit radix passes
// over the row (read from memory each pass) per order statistic, on an order-preserving integer image of the floats.
__device__ __forceinline__ unsigned rq_key(float x) {
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float rq_val(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

__global__ __launch_bounds__(256) void row_mean_quantiles_select_kernel(const float *in, long long n_rows, int m, double q_lo,
                                                                       double q_hi, float *mean, float *lo, float *hi) {
  __shared__ double red[256];
  __shared__ unsigned hist[256];
  __shared__ unsigned sel_prefix, sel_rank;
  for (long long row = blockIdx.x; row < n_rows; row += gridDim.x) {
    const float *src = in + row * (long long)m;
    double s = 0.0;
    for (int i = threadIdx.x; i < m; i += 256) s += (double)src[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
      __syncthreads();
    }
    const double qs[2] = {q_lo, q_hi};
    float res[2];
    for (int t = 0; t < 2; ++t) {
      const double vi = qs[t] * (double)(m - 1);
      int i0 = (int)floor(vi);
      i0 = i0 < 0 ? 0 : (i0 > m - 1 ? m - 1 : i0);
      const int i1 = i0 + 1 > m - 1 ? m - 1 : i0 + 1;
      float v01[2];
      for (int w = 0; w < 2; ++w) {
        if (w == 1 && i1 == i0) { v01[1] = v01[0]; break; }
        unsigned prefix = 0u, mask = 0u;
        if (threadIdx.x == 0) sel_rank = (unsigned)(w ? i1 : i0);
        for (int shift = 24; shift >= 0; shift -= 8) {
          hist[threadIdx.x] = 0u;
          __syncthreads();
          for (int i = threadIdx.x; i < m; i += 256) {
            const unsigned k = rq_key(src[i]);
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
          }
          __syncthreads();
          if (threadIdx.x == 0) {
            unsigned r = sel_rank, b = 0u;
            while (r >= hist[b]) { r -= hist[b]; ++b; }
            sel_rank = r;
            sel_prefix = prefix | (b << shift);
          }
          __syncthreads();
          prefix = sel_prefix;
          mask |= 255u << shift;
        }
        v01[w] = rq_val(prefix);
      }
      res[t] = np_lerp(v01[0], v01[1], vi - (double)i0);
    }
    if (threadIdx.x == 0) {
      mean[row] = (float)(red[0] / (double)m);
      lo[row] = res[0];
      hi[row] = res[1];
    }
    __syncthreads();
  }
}

extern "C" int bgm_row_mean_quantiles(bgm_handle *h, const float *in, int64_t n_rows, int32_t m, double q_lo,
                                      double q_hi, float *mean, float *lo, float *hi, void *stream_) {
  if (!h || !in || !mean || !lo || !hi || m <= 0 || q_lo < 0 || q_lo > 1 || q_hi < 0 || q_hi > 1) {
    bgm_set_error("bgm_row_mean_quantiles: bad argument"); return BGM_E_INVALID;
  }
  if (n_rows <= 0) return BGM_OK;
  int m_pow2 = 1;
  while (m_pow2 < m) m_pow2 <<= 1;
  if (m_pow2 < 2) m_pow2 = 2;
  BGM_HIP_CHECK(hipSetDevice(h->device));
  if (m_pow2 > 32768) {          // the row does not fit one workgroup's LDS sort: radix selection of the order statistics
    const int grid = (int)std::min<int64_t>(n_rows, (int64_t)h->n_cus * 8);
    hipLaunchKernelGGL(row_mean_quantiles_select_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_, in, (long long)n_rows, m,
                       q_lo, q_hi, mean, lo, hi);
    BGM_HIP_CHECK(hipGetLastError());
    return BGM_OK;
  }
  const int lds = m_pow2 * 4;
  BGM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(row_mean_quantiles_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int grid = (int)std::min<int64_t>(n_rows, (int64_t)h->n_cus * 8);
  hipLaunchKernelGGL(row_mean_quantiles_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream_, in,
                     (long long)n_rows, m, m_pow2, q_lo, q_hi, mean, lo, hi);
  BGM_HIP_CHECK(hipGetLastError());
  return BGM_OK;
}
