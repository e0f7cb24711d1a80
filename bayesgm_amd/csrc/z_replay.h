// z_replay.h -- replay of the pending zero-gradient Adam steps of latent rows (shared by the deterministic and the Bayesian fit paths).
// Keras applies the latent optimizer's sparse gradient with dense moment decay (causalbgm/base.py:246-302 through
// tf.keras.optimizers.Adam._resource_apply_sparse): EVERY row of the [N x q] table moves at every minibatch.
#pragma once
#include <hip/hip_runtime.h>

// Deferred form of the dense-decay latent Adam ("replay", lazy = 2; fit_adam_z_kernel mode 0, bnn_z_decay / bnn_z_apply).  For a row outside the minibatch the dense-decay step has g = 0:
//   m <- b1 m,  v <- b2 v,  z <- z - lr_t m / (sqrt(v) + eps),
// so k pending steps after step t0 are m b1^k, v b2^k and z - sum_{j=1..k} lr_{t0+j} m b1^j / (sqrt(v b2^j) + eps), a series whose
// terms shrink by b1 / sqrt(b2) = 0.9045 (faster once sqrt(v) < eps): FIT_REPLAY_TERMS of them carry it to 7e-12 of its first term.
// t_last[row] = the step the row's (z, m, v) are current to; idx = NULL replays every row (flush).  16 lanes share one element:
// lane l starts from the closed form at term 16 l' and walks its chunk, the partial sums meet through DPP shuffles.
#define FIT_REPLAY_TERMS 256
__device__ __forceinline__ float fit_adam_lr_t(float lr, float t, float ln_b1, float ln_b2) {
  return lr * sqrtf(-expm1f(t * ln_b2)) / (-expm1f(t * ln_b1));          // lr sqrt(1 - b2^t) / (1 - b1^t) without cancellation
}

__device__ __forceinline__ void fit_adam_z_replay_one(long long gid, int l, float *z, float *zm, float *zv, const int *t_last, int q,
                                                      const int *idx, long long n_sel, int t_to, float lr, float b1, float b2, float eps) {
  const long long el = gid >> 4;
  if (el >= n_sel * q) return;                     // (whole 16-lane groups leave together)
  const long long s = el / q;
  const int f = (int)(el - s * q);
  const long long row = idx ? (long long)idx[s] : s;
  const int t0 = t_last[row];
  const int k = t_to - t0;
  if (k <= 0) return;
  const long long e = row * q + f;
  const float m = zm[e], v = zv[e];
  const float l2b1 = log2f(b1), l2b2 = log2f(b2), ln_b1 = logf(b1), ln_b2 = logf(b2);
  const int J = k < FIT_REPLAY_TERMS ? k : FIT_REPLAY_TERMS;
  const int C = (J + 15) >> 4;
  const int j0 = l * C;
  float mj = m * exp2f((float)j0 * l2b1), vj = v * exp2f((float)j0 * l2b2), sum = 0.0f;
  for (int c = 1; c <= C; ++c) {
    mj *= b1; vj *= b2;
    if (j0 + c <= J) sum += fit_adam_lr_t(lr, (float)(t0 + j0 + c), ln_b1, ln_b2) * mj / (sqrtf(vj) + eps);
  }
#pragma unroll
  for (int o = 8; o; o >>= 1) sum += __shfl_xor(sum, o, 16);
  if (l == 0) {
    z[e] -= sum;
    zm[e] = m * exp2f((float)k * l2b1);
    zv[e] = v * exp2f((float)k * l2b2);
  }
}

static __global__ void fit_adam_z_replay_kernel(float *z, float *zm, float *zv, const int *t_last, int q, const int *idx, long long n_sel,
                                         int t_to, float lr, float b1, float b2, float eps) {
  fit_adam_z_replay_one((long long)blockIdx.x * blockDim.x + threadIdx.x, (int)(threadIdx.x & 15), z, zm, zv, t_last, q, idx, n_sel, t_to, lr, b1, b2,
                        eps);
}

// After a replay of the rows idx[0..n_sel) their (z, m, v) are current to `value`: recorded in a launch of its own (the replay kernel's
// threads of one row read t_last[row] concurrently), so that a second replay of the same rows -- a flush after an interrupted
// minibatch -- finds nothing pending.
static __global__ void fit_mark_rows_kernel(int *t_last, const int *idx, long long n_sel, int value) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_sel) t_last[idx[i]] = value;
}

static __global__ void fit_fill_int_kernel(int *a, long long n, int value) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = value;
}

