// causal_event_kernels.h -- the retained ("keep") phase of CausalBGM.predict with the outcome net taken out of the sampler (gfx950).
//
// replaces (reference, src/bayesgm/models/causalbgm/base.py): the retained iterations of metropolis_hastings_sampler :860-899 together
// with infer_from_latent_posterior :671-763 (the f-net at every dose for every retained draw of every row).
//
// A retained draw of a Metropolis-Hastings chain equals the previous one unless the proposal was accepted (base.py:868-871), and the
// outcome net f is a deterministic function of the chain state: per CHAIN only a fraction a (the acceptance rate, 2-10 % at the
// reference's q_sd = 1) of the retained draws needs f at all.  The fused sampler (causal_mh_kernel<EFFECT = 1>) can only skip f for a
// whole wave of 16 chains, i.e. with probability (1 - a)^16.  Here the phase is three kernels per segment of S retained iterations:
//   1. causal_mh_kernel<EFFECT = 3>  -- the pure-transition kernel; an accepted move appends an EVENT (chain of the tile, iteration,
//      new state) to the wave slot's region, in time order (every chain emits one at the first iteration of a call);
//   2. causal_event_f_kernel        -- f at all doses on DENSE 16-event tiles: the slot's events are contiguous, so every MFMA column
//      is a state that needs evaluating; writes (mean, sd) of every (event, dose) in the lane layout of causal_effects' cache;
//   3. causal_event_spread_kernel   -- per row tile, in the sampler's slot order: walks the retained iterations, picks up the tile's
//      events as they fall due, and adds  mean + sd * noise(row, iteration, dose)  to the slot's ADRF sums with the same Philox
//      calls, the same fma, the same 16-row reduction and the same slot-private atomics as causal_effects / causal_effects_cached.
// A column of a 16x16 MFMA depends on that column's operands only, so (mean, sd) of a state are the bits the fused kernel computes
// for it, and the sums are accumulated per (slot, draw, dose) in the same tile order: the ADRF is bit-identical to the fused path.
#pragma once
#include "causal_kernels.h"

#define EV_NCMAX 8      // Philox calls (= passes of four doses) per retained draw held in registers: up to 32 doses

struct CausalEventFArgs {
  const float *blob;
  const float *ev_z;
  const int *slot_cnt;
  long long ev_cap;
  int n_doses;
  const float *x_values;
  float2 *ev_out;                    // [n_slots * ev_cap / 16][n_calls][64]: (mean, sd) of the dose lane group g finishes in that pass
  unsigned long long *eff_stats;     // [1] += events evaluated
  CausalMeta m;
};

template <int KT1, int KSL1, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void causal_event_f_kernel(CausalEventFArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const CausalMeta &m = a.m;
  lds_fill(lds, a.blob, m.total);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4, lane_off = 64 * g + j;
  const long long slot = (long long)blockIdx.x * WAVES + wave;
  const int cnt = a.slot_cnt[slot];
  const int n_calls = (a.n_doses + 3) >> 2;
  if (a.eff_stats != nullptr && lane == 0 && cnt > 0) atomicAdd(&a.eff_stats[1], (unsigned long long)cnt);
  for (int e0 = 0; e0 < cnt; e0 += 16) {
    BGM_NO_HOIST();
    const int el = (e0 + j < cnt) ? e0 + j : cnt - 1;           // a partly filled last tile repeats the last event (its columns are not read)
    const float *zr = a.ev_z + (slot * a.ev_cap + el) * (long long)m.q;
    f32x4 zs[1][KT1];
#pragma unroll
    for (int t = 0; t < KT1; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 16 * t + 4 * r + g;
        zs[0][t][r] = (f < m.q) ? zr[f] : 0.0f;
      }
    const unsigned rowid[1] = {0u};
    const bool valid[1] = {true};
    float2 *out = a.ev_out + ((slot * a.ev_cap + e0) >> 4) * (long long)n_calls * 64;
    causal_effects<KT1, KSL1, 1, 1, true, true, true>(lds, m, lane_off, g, j, lane, zs, rowid, valid, 0ll, 16ll, 0u, 0ll, 1, 0, a.n_doses,
                                                      a.x_values, nullptr, nullptr, 0u, 0u, out);
  }
}

struct CausalEventSpreadArgs {
  long long n, row_base;
  int it_begin, n_iters, burn_in, n_keep;
  int sample_y, n_doses;
  unsigned k0, k1;
  const unsigned *ev_meta;
  const int *tile_ev;
  long long ev_cap;
  const float2 *ev_out;
  float2 *carry;                     // [n_tiles][n_calls][64]: (mean, sd) of every chain's current state between the segments of a call
  int first;                         // 1: the segment starts the call (every chain has an event at its first iteration; carry is not read)
  float *adrf_partial;               // [n_slots][n_keep][n_doses]
};

// One wave per sampler slot (the same slot -> tile mapping and order as causal_mh_kernel), WAVES waves per block, no LDS.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void causal_event_spread_kernel(CausalEventSpreadArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const long long n = a.n, n_tiles = (n + 15) / 16;
  const long long slot = (long long)blockIdx.x * WAVES + wave, n_slots = (long long)gridDim.x * WAVES;
  const int nd = a.n_doses, n_calls = (nd + 3) >> 2, n_own = n_calls & ~3;
  float *adrf_slot = a.adrf_partial + slot * (long long)nd * a.n_keep;
  for (long long tile = slot; tile < n_tiles; tile += n_slots) {
    const long long row = tile * 16 + j;
    const bool valid = row < n;
    const unsigned rowid = (unsigned)(a.row_base + (valid ? row : n - 1));
    float2 cur[EV_NCMAX];
#pragma unroll
    for (int kb = 0; kb < EV_NCMAX; ++kb) cur[kb] = make_float2(0.0f, 0.0f);
    float2 *cr = a.carry + tile * (long long)n_calls * 64 + lane;
    if (!a.first) {
#pragma unroll
      for (int kb = 0; kb < EV_NCMAX; ++kb)
        if (kb < n_calls) cur[kb] = cr[kb * 64];
    }
    const int eb = a.tile_ev[2 * tile], ec = a.tile_ev[2 * tile + 1];
    const long long base = slot * a.ev_cap + eb;      // the tile's first event
    // the tile's event words, 64 at a time, one per lane
    int p = 0, chunk0 = 0;
    unsigned mchunk = (lane < ec) ? a.ev_meta[base + lane] : 0xFFFFFFFFu;
    for (int dl = 0; dl < a.n_iters; ++dl) {
      const unsigned it = (unsigned)(a.it_begin + dl);
      const long long d = (long long)it - a.burn_in;
      // ---- events falling due at this iteration: the chain's lanes request their new pairs (consumed behind the noise below)
      int mine = -1;
      while (p < ec) {
        if (p - chunk0 >= 64) { chunk0 = p; mchunk = (p + lane < ec) ? a.ev_meta[base + p + lane] : 0xFFFFFFFFu; }
        const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)mchunk, __builtin_amdgcn_readfirstlane(p - chunk0));
        if ((int)(w >> 4) != dl) break;
        mine = ((int)(w & 15u) == j) ? p : mine;
        ++p;
      }
      float2 nxt[EV_NCMAX];
#pragma unroll
      for (int kb = 0; kb < EV_NCMAX; ++kb) nxt[kb] = cur[kb];
      if (mine >= 0) {
        const long long e = base + mine;
        const float2 *o = a.ev_out + (e >> 4) * (long long)n_calls * 64 + 16 * g + (int)(e & 15);
#pragma unroll
        for (int kb = 0; kb < EV_NCMAX; ++kb)
          if (kb < n_calls) nxt[kb] = o[kb * 64];
      }
      // ---- outcome noise of (row, iteration): lane group g evaluates call 4c + g of a group of four calls once and uses its word p
      //      in pass 4c + p (dose 16c + 4g + p); a remainder of fewer than four calls is evaluated by every lane group (word g)
      f32x4 nz_own[EV_NCMAX / 4];
#pragma unroll
      for (int c = 0; c < EV_NCMAX / 4; ++c) {
        nz_own[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (a.sample_y && 4 * c < n_own) nz_own[c] = box_muller4(philox4x32_10(rowid, it, (unsigned)(4 * c + g), TAG_YNOISE, a.k0, a.k1));
      }
#pragma unroll
      for (int kb = 0; kb < EV_NCMAX; ++kb) cur[kb] = nxt[kb];
      // ---- the draw's contribution: same expression, reduction and atomics as causal_effects' grouped path
#pragma unroll
      for (int kb = 0; kb < EV_NCMAX; ++kb) {
        if (kb < n_calls) {
          const bool own = kb < n_own;
          const int k = own ? 4 * ((kb & ~3) + g) + (kb & 3) : 4 * kb + g;
          float noise = nz_own[kb >> 2][kb & 3];
          if (!own && a.sample_y) {
            const f32x4 zh = box_muller4(philox4x32_10(rowid, it, (unsigned)kb, TAG_YNOISE, a.k0, a.k1));
            noise = pick_by_group(g, zh[0], zh[1], zh[2], zh[3]);
          }
          float y = a.sample_y ? fmaf(cur[kb].y, noise, cur[kb].x) : cur[kb].x;
          y = (valid && k < nd) ? y : 0.0f;
          const float tot = sum_over_j_to_lane15(y);
          if (j == 15 && k < nd) unsafeAtomicAdd(adrf_slot + d * nd + k, tot);
        }
      }
    }
#pragma unroll
    for (int kb = 0; kb < EV_NCMAX; ++kb)
      if (kb < n_calls) cr[kb * 64] = cur[kb];
  }
}
