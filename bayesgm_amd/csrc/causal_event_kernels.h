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

// Only the outcome net's weights go to LDS (16-20 KB of the ~150 KB sampling blob: first layer, x row, the 64 -> 32 -> 8 -> 2 tail), so
// that four 4-wave workgroups share a CU instead of one 8-wave workgroup: the dose passes are short dependent MFMA chains, and it is
// other waves that fill their gaps.  WPS waves deal a slot's event tiles among themselves.
template <int KT1, int KSL1, int WAVES, int WPS>
__global__ __launch_bounds__(64 * WAVES) void causal_event_f_kernel(CausalEventFArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  CausalMeta m = a.m;
  {
    int off = 0;
    auto put = [&](int src, int nfl) {
      for (int i = threadIdx.x; i < nfl; i += 64 * WAVES) lds[off + i] = a.blob[src + i];
      const int o = off;
      off += nfl;
      return o;
    };
    m.w1f = put(a.m.w1f, 16 * KT1 * 64); m.b1f = put(a.m.b1f, 64);
    m.wf2 = put(a.m.wf2, 64 * 32); m.bf2 = put(a.m.bf2, 32); m.wf3 = put(a.m.wf3, 32 * 16); m.bf3 = put(a.m.bf3, 16);
    m.wf4 = put(a.m.wf4, 16 * 16); m.bf4 = put(a.m.bf4, 16); m.wxf = put(a.m.wxf, 64);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4, lane_off = 64 * g + j;
  const long long wid = (long long)blockIdx.x * WAVES + wave;
  const long long slot = wid / WPS;
  const int sub = (int)(wid % WPS);
  const int cnt = a.slot_cnt[slot];
  const int n_calls = (a.n_doses + 3) >> 2;
  if (a.eff_stats != nullptr && lane == 0 && sub == 0 && cnt > 0) atomicAdd(&a.eff_stats[1], (unsigned long long)cnt);
  for (int e0 = 16 * sub; e0 < cnt; e0 += 16 * WPS) {
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

// ---- binary treatment: individual treatment effects (round 6).  The same three steps; the outcome net is evaluated at the two arms
// x = 1, x = 0 of every EVENT (causal_effects<EFFECT = 2> with the noise and the store compiled out: the (mean, sd) pairs the fused
// kernel forms for that state, bit for bit), 16 bytes per event; the spread is one thread per chain, which walks its tile's events in
// time order and writes  y(1) - y(0) = fma(sd1, e0, mean1) - fma(sd0, e1, mean0)  of every retained draw with the noise words of
// Philox(row, iteration, 0, TAG_YNOISE) -- the expression of causal_effects / causal_ite_cached, so the ITE matrix equals the fused
// kernel's to the last bit.  replaces: infer_from_latent_posterior's binary branch, causalbgm/base.py:686-733.
template <int KT1, int KSL1, int WAVES, int WPS>
__global__ __launch_bounds__(64 * WAVES) void causal_event_f_ite_kernel(CausalEventFArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  CausalMeta m = a.m;
  {
    int off = 0;
    auto put = [&](int src, int nfl) {
      for (int i = threadIdx.x; i < nfl; i += 64 * WAVES) lds[off + i] = a.blob[src + i];
      const int o = off;
      off += nfl;
      return o;
    };
    m.w1f = put(a.m.w1f, 16 * KT1 * 64); m.b1f = put(a.m.b1f, 64);
    m.wf2 = put(a.m.wf2, 64 * 32); m.bf2 = put(a.m.bf2, 32); m.wf3 = put(a.m.wf3, 32 * 16); m.bf3 = put(a.m.bf3, 16);
    m.wf4 = put(a.m.wf4, 16 * 16); m.bf4 = put(a.m.bf4, 16); m.wxf = put(a.m.wxf, 64);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4, lane_off = 64 * g + j;
  const long long wid = (long long)blockIdx.x * WAVES + wave;
  const long long slot = wid / WPS;
  const int sub = (int)(wid % WPS);
  const int cnt = a.slot_cnt[slot];
  if (a.eff_stats != nullptr && lane == 0 && sub == 0 && cnt > 0) atomicAdd(&a.eff_stats[1], (unsigned long long)cnt);
  float4 *out4 = reinterpret_cast<float4 *>(a.ev_out);
  for (int e0 = 16 * sub; e0 < cnt; e0 += 16 * WPS) {
    BGM_NO_HOIST();
    const int el = (e0 + j < cnt) ? e0 + j : cnt - 1;
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
    float c[4];
    causal_effects<KT1, KSL1, 1, 2, true, true, true>(lds, m, lane_off, g, j, lane, zs, rowid, valid, 0ll, 16ll, 0u, 0ll, 1, 0, 2,
                                                      nullptr, nullptr, nullptr, 0u, 0u, nullptr, c);
    if (g == 0 && e0 + j < cnt) out4[slot * a.ev_cap + e0 + j] = make_float4(c[0], c[1], c[2], c[3]);      // (mean, sd) of arm 1, of arm 0
  }
}

struct CausalEventIteArgs {
  long long n, row_base;
  int it_begin, n_iters, burn_in, n_keep;
  int sample_y, n_slots;
  unsigned k0, k1;
  const unsigned *ev_meta;
  const int *tile_ev;
  long long ev_cap;
  const float4 *ev_out;              // per event: (mean1, sd1, mean0, sd0)
  const float4 *carry_in;            // [n]: every chain's pairs at the end of the previous segment of the call
  float4 *carry_out;
  int first;
  float *ite;                        // [n x n_keep]
};

static __global__ __launch_bounds__(256) void causal_event_ite_kernel(CausalEventIteArgs a) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  if (row >= a.n) return;
  const long long tile = row >> 4, slot = tile % a.n_slots;      // (the sampler deals tile t to slot t mod n_slots)
  const int j = (int)(row & 15);
  const int eb = a.tile_ev[2 * tile], ec = a.tile_ev[2 * tile + 1];
  const long long base = slot * a.ev_cap + eb;
  const unsigned rowid = (unsigned)(a.row_base + row);
  float4 cur = a.first ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : a.carry_in[row];
  int p = 0;
  unsigned w = p < ec ? a.ev_meta[base] : 0xFFFFFFFFu;
  for (int dl = 0; dl < a.n_iters; ++dl) {
    while (p < ec && (int)(w >> 4) <= dl) {                       // the tile's events up to this iteration, in time order
      if ((int)(w & 15u) == j) cur = a.ev_out[base + p];
      ++p;
      w = p < ec ? a.ev_meta[base + p] : 0xFFFFFFFFu;
    }
    const unsigned it = (unsigned)(a.it_begin + dl);
    f32x4 nz = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (a.sample_y) nz = box_muller4(philox4x32_10(rowid, it, 0u, TAG_YNOISE, a.k0, a.k1));
    const float y1 = a.sample_y ? fmaf(cur.y, nz[0], cur.x) : cur.x;
    const float y0 = a.sample_y ? fmaf(cur.w, nz[1], cur.z) : cur.z;
    a.ite[row * (long long)a.n_keep + ((long long)it - a.burn_in)] = y1 - y0;
  }
  a.carry_out[row] = cur;
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
  const float2 *carry_in;            // [n_tiles][n_calls][64]: (mean, sd) of every chain's state at the end of the previous segment of the call
  float2 *carry_out;                 // the same at the end of this one (another buffer: the block's waves read carry_in at different times)
  int first;                         // 1: the segment starts the call (every chain has an event at its first iteration; carry_in is not read)
  float *adrf_partial;               // [n_slots][n_keep][n_doses]
};

// One BLOCK of SW waves per sampler slot (the same slot -> tile mapping and tile order as causal_mh_kernel).  The segment's retained
// iterations are dealt to the block's waves in contiguous sub-ranges: the sums of different draws are different words, so every
// word still receives its tiles' contributions in tile order (what makes the result independent of the split, and equal to the
// fused kernel's), while the slot has SW times the waves in flight to cover the latency of the event loads.  A wave finds its
// chains' states at the start of its sub-range by walking the tile's earlier events (event words only) and loading the pairs of the
// last one per chain (the carried pairs if there is none).  The sums are accumulated in LDS ([iterations of the segment][doses],
// LDS float adds: the same fp32 addition, in the same order, as the fused kernel's slot-private L2 atomics) and added to the slot's
// partial sums once per segment with plain coalesced stores.
template <int SW>
__global__ __launch_bounds__(64 * SW) void causal_event_spread_kernel(CausalEventSpreadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float acc_lds[];      // [n_iters][n_doses]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const long long n = a.n, n_tiles = (n + 15) / 16;
  const long long slot = blockIdx.x, n_slots = gridDim.x;
  const int nd = a.n_doses, n_calls = (nd + 3) >> 2, n_own = n_calls & ~3;
  for (int i = threadIdx.x; i < a.n_iters * nd; i += 64 * SW) acc_lds[i] = 0.0f;
  __syncthreads();
  // this wave's iterations of the segment
  const int per = (a.n_iters + SW - 1) / SW;
  const int dl0 = wave * per < a.n_iters ? wave * per : a.n_iters, dl1 = dl0 + per < a.n_iters ? dl0 + per : a.n_iters;
  for (long long tile = slot; tile < n_tiles; tile += n_slots) {
    const long long row = tile * 16 + j;
    const bool valid = row < n;
    const unsigned rowid = (unsigned)(a.row_base + (valid ? row : n - 1));
    const int eb = a.tile_ev[2 * tile], ec = a.tile_ev[2 * tile + 1];
    const long long base = slot * a.ev_cap + eb;      // the tile's first event
    // the tile's event words, 64 at a time, one per lane
    int p = 0, chunk0 = 0;
    unsigned mchunk = (lane < ec) ? a.ev_meta[base + lane] : 0xFFFFFFFFu;
    auto word = [&](int pp) -> unsigned {
      if (pp - chunk0 >= 64) { chunk0 = pp; mchunk = (pp + lane < ec) ? a.ev_meta[base + pp + lane] : 0xFFFFFFFFu; }
      return (unsigned)__builtin_amdgcn_readlane((int)mchunk, __builtin_amdgcn_readfirstlane(pp - chunk0));
    };
    auto pairs_of = [&](int ev_idx, float2 (&dst)[EV_NCMAX]) {
      const long long e = base + ev_idx;
      const float2 *o = a.ev_out + (e >> 4) * (long long)n_calls * 64 + 16 * g + (int)(e & 15);
#pragma unroll
      for (int kb = 0; kb < EV_NCMAX; ++kb)
        if (kb < n_calls) dst[kb] = o[kb * 64];
    };
    // ---- the chains' pairs at iteration dl0: the last event before it, else what the previous segment left
    float2 cur[EV_NCMAX];
#pragma unroll
    for (int kb = 0; kb < EV_NCMAX; ++kb) cur[kb] = make_float2(0.0f, 0.0f);
    const float2 *cr = a.carry_in + tile * (long long)n_calls * 64 + lane;
    {
      int last = -1;
      while (p < ec) {
        const unsigned w = word(p);
        if ((int)(w >> 4) >= dl0) break;
        last = ((int)(w & 15u) == j) ? p : last;
        ++p;
      }
      if (last >= 0) pairs_of(last, cur);
      else if (!a.first) {
#pragma unroll
        for (int kb = 0; kb < EV_NCMAX; ++kb)
          if (kb < n_calls) cur[kb] = cr[kb * 64];
      }
    }
    for (int dl = dl0; dl < dl1; ++dl) {
      const unsigned it = (unsigned)(a.it_begin + dl);
      // ---- events falling due at this iteration: the chain's lanes request their new pairs (consumed behind the noise below)
      int mine = -1;
      while (p < ec) {
        const unsigned w = word(p);
        if ((int)(w >> 4) != dl) break;
        mine = ((int)(w & 15u) == j) ? p : mine;
        ++p;
      }
      float2 nxt[EV_NCMAX];
#pragma unroll
      for (int kb = 0; kb < EV_NCMAX; ++kb) nxt[kb] = cur[kb];
      if (mine >= 0) pairs_of(mine, nxt);
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
      // ---- the draw's contribution: same expression and 16-row reduction as causal_effects' grouped path
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
          if (j == 15 && k < nd) atomicAdd(&acc_lds[dl * nd + k], tot);      // (ds_add_f32: this wave owns the word)
        }
      }
    }
    if (dl1 == a.n_iters && dl0 < dl1) {      // the wave holding the segment's last iteration leaves the pairs for the next segment
      float2 *co = a.carry_out + tile * (long long)n_calls * 64 + lane;
#pragma unroll
      for (int kb = 0; kb < EV_NCMAX; ++kb)
        if (kb < n_calls) co[kb * 64] = cur[kb];
    }
  }
  __syncthreads();
  float *adrf_slot = a.adrf_partial + slot * (long long)nd * a.n_keep + ((long long)a.it_begin - a.burn_in) * nd;
  for (int i = threadIdx.x; i < a.n_iters * nd; i += 64 * SW) adrf_slot[i] += acc_lds[i];
}
