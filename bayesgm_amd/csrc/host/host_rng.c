/* host_rng.c -- the host random numbers of one block of EGM iterations, drawn from NumPy's legacy global generator state in the
 * reference's order (causalbgm/base.py:399-416: per iteration g_d_freq x [np.random.choice(n, B, replace=False), z ~ N(0, 1) [B x q],
 * eps ~ U(0, 1)], then z and the indices of the generator step).
 *
 * Why C: np.random.choice(n, B, replace=False) permutes all n indices per call; at the tutorial's n = 20 000 that is ~0.28 ms of
 * interpreter + generic-shuffle time, 180 000 times per warm start -- 30-50 s, more than the whole GPU side of the fit (7 s) once
 * the step kernels were rebuilt.  The same Fisher-Yates sweep on the same MT19937 stream in a tight loop is ~3x faster and
 * bit-identical (tests/test_host_rng.py checks every output and the generator state against NumPy).
 * Algorithms restated from NumPy's legacy RandomState semantics: MT19937 tempering, random_interval (masked rejection),
 * legacy_double (53-bit from two draws), legacy_gauss (polar method with one cached value). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t key[624]; uint32_t out[624]; int pos; int has_gauss; double gauss; } bgm_mt_state;   /* out: key, tempered */

static void mt_gen(bgm_mt_state *s) {
  const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
  uint32_t y;
  int i;
  for (i = 0; i < 624 - 397; i++) { y = (s->key[i] & UPPER) | (s->key[i + 1] & LOWER); s->key[i] = s->key[i + 397] ^ (y >> 1) ^ (-(y & 1) & MATRIX_A); }
  for (; i < 623; i++) { y = (s->key[i] & UPPER) | (s->key[i + 1] & LOWER); s->key[i] = s->key[i + (397 - 624)] ^ (y >> 1) ^ (-(y & 1) & MATRIX_A); }
  y = (s->key[623] & UPPER) | (s->key[0] & LOWER);
  s->key[623] = s->key[396] ^ (y >> 1) ^ (-(y & 1) & MATRIX_A);
  s->pos = 0;
}
static void mt_temper(bgm_mt_state *s) {          /* the whole block at once: a vectorisable loop instead of four dependent ops per draw */
  for (int i = 0; i < 624; i++) {
    uint32_t y = s->key[i];
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    s->out[i] = y;
  }
}
static inline uint32_t mt_u32(bgm_mt_state *s) {
  if (__builtin_expect(s->pos == 624, 0)) { mt_gen(s); mt_temper(s); }
  return s->out[s->pos++];
}
static inline double mt_double(bgm_mt_state *s) {
  const int32_t a = (int32_t)(mt_u32(s) >> 5), b = (int32_t)(mt_u32(s) >> 6);
  return (a * 67108864.0 + b) / 9007199254740992.0;
}
static inline uint32_t mt_interval(bgm_mt_state *s, uint32_t max) {       /* uniform on [0, max], max < 2^32 */
  uint32_t mask = max, v;
  if (max == 0) return 0;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  while ((v = (mt_u32(s) & mask)) > max) {}
  return v;
}
static inline double mt_gauss(bgm_mt_state *s) {
  if (s->has_gauss) { const double t = s->gauss; s->has_gauss = 0; s->gauss = 0.0; return t; }
  double f, x1, x2, r2;
  do { x1 = 2.0 * mt_double(s) - 1.0; x2 = 2.0 * mt_double(s) - 1.0; r2 = x1 * x1 + x2 * x2; } while (r2 >= 1.0 || r2 == 0.0);
  f = sqrt(-2.0 * log(r2) / r2);
  s->gauss = f * x1; s->has_gauss = 1;
  return f * x2;
}
/* np.random.choice(n, k, replace=False): permutation(n)[:k] */
static void mt_choice(bgm_mt_state *s, int32_t *perm, int n, int k, int32_t *out) {
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int i = n - 1; i >= 1; --i) { const uint32_t jj = mt_interval(s, (uint32_t)i); const int32_t t = perm[i]; perm[i] = perm[jj]; perm[jj] = t; }
  for (int i = 0; i < k; ++i) out[i] = perm[i];
}

/* key[624], pos, has_gauss, gauss: in/out (np.random.get_state(legacy=True) / set_state).
 * idx [n_it][steps][B], z [n_it][steps][B][q] (float32), eps [n_it][g_d_freq][n_eps] with steps = g_d_freq + 1.
 * order per iteration: g_d_freq x (choice, z, n_eps uniforms), then z, choice   -- choice_first_in_gen = 0 (CausalBGM)
 * Returns 0, or -1 on a bad argument. */
int bgm_host_egm_block(uint32_t *key, int *pos, int *has_gauss, double *gauss, int n, int B, int q, int n_it, int g_d_freq, int n_eps,
                       int32_t *idx, float *z, double *eps) {
  if (!key || !pos || !has_gauss || !gauss || n < 1 || B < 1 || B > n || q < 1 || n_it < 0 || g_d_freq < 0 || n_eps < 0) return -1;
  bgm_mt_state s;
  memcpy(s.key, key, sizeof(s.key));
  s.pos = *pos; s.has_gauss = *has_gauss; s.gauss = *gauss;
  mt_temper(&s);
  int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  if (!perm) return -1;
  const int steps = g_d_freq + 1;
  for (int it = 0; it < n_it; ++it) {
    for (int j = 0; j < g_d_freq; ++j) {
      mt_choice(&s, perm, n, B, idx + ((size_t)it * steps + j) * B);
      float *zz = z + ((size_t)it * steps + j) * B * q;
      for (int e = 0; e < B * q; ++e) zz[e] = (float)(0.0 + 1.0 * mt_gauss(&s));
      for (int e = 0; e < n_eps; ++e) eps[((size_t)it * g_d_freq + j) * n_eps + e] = 0.0 + 1.0 * mt_double(&s);
    }
    float *zz = z + ((size_t)it * steps + g_d_freq) * B * q;
    for (int e = 0; e < B * q; ++e) zz[e] = (float)(0.0 + 1.0 * mt_gauss(&s));
    mt_choice(&s, perm, n, B, idx + ((size_t)it * steps + g_d_freq) * B);
  }
  free(perm);
  memcpy(key, s.key, sizeof(s.key));
  *pos = s.pos; *has_gauss = s.has_gauss; *gauss = s.gauss;
  return 0;
}
