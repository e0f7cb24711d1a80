// fit_sync.h -- ordering of kernels on two HIP streams through counters in device memory (no events on the critical path).
#pragma once
#include <hip/hip_runtime.h>

// Device-side ordering between the two streams of bgm_causal_fit_epoch / bgm_bnn_fit_epoch: a kernel may spin at entry until a counter in device memory
// reaches a target, and every workgroup of a kernel may add 1 to a counter at its end.  The producer of a wait is always issued
// before its consumer (host order), both grids are a handful of workgroups, so the producer is never starved; the spin is bounded all
// the same (FIT_SYNC_TIMEOUT_TICKS of the 100 MHz wall clock, then *err = 1 and every later wait falls through).
struct FitSync {
  unsigned *wait_ctr; unsigned wait_target;
  unsigned *done_ctr;
  int *err;
};
#define FIT_SYNC_TIMEOUT_TICKS 2000000ull      // 20 ms
__device__ __forceinline__ void fit_sync_wait(const FitSync &s) {
  if (!s.wait_ctr) return;
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(s.wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - s.wait_target) < 0) {
      if (__hip_atomic_load(s.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > FIT_SYNC_TIMEOUT_TICKS) { __hip_atomic_store(s.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // what the producer released is visible to every thread from here on
}
__device__ __forceinline__ void fit_sync_done(const FitSync &s) {
  if (!s.done_ctr) return;
  __syncthreads();                 // (workgroup-scope release: every wave's stores have reached this XCD's L2)
  if (threadIdx.x == 0) {          // one write-back of that L2 per workgroup, not one per wave: the L2 sweep is what costs
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(s.done_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
