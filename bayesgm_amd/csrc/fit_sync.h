// fit_sync.h -- ordering of kernels on two HIP streams through counters in device memory (no events on the critical path).
#pragma once
#include <hip/hip_runtime.h>
#include <ctime>

// Device-side ordering between the two streams of bgm_causal_fit_epoch / bgm_bnn_fit_epoch: a kernel may spin at entry until a counter in device memory
// reaches a target, and every workgroup of a kernel may add 1 to a counter at its end.  The producer of a wait is always issued
// before its consumer (host order), both grids are a handful of workgroups, so the producer is never starved; the spin is bounded all
// the same (FIT_SYNC_TIMEOUT_TICKS of the 100 MHz wall clock, then *err = 1 and every later wait falls through; the epoch call reads
// the word behind its join and fails).
struct FitSync {
  unsigned *wait_ctr; unsigned wait_target;
  unsigned *done_ctr;
  int *err;
};
#define FIT_SYNC_TIMEOUT_TICKS 200000000ull    // 2 s: a GPU shared with another process or a debugger attach must not void an epoch
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

// Capability check, once per handle: do kernels of the two streams really run side by side?  The ordering above relies on it -- under a
// tool that lets one kernel run at a time (rocprofv3 --pmc serialises dispatches and picks among the ready queues in no particular
// order), or if the runtime mapped both streams to one hardware queue, a consumer dispatched before its producer would spin until
// the bound.  The probe is the hostile case itself: a waiter is submitted FIRST on one stream, its poster afterwards on the other.
// probe word: 1 = the waiter saw the post, 2 = it gave up (2 ms): the caller then orders the streams with HIP events.
static __global__ void fit_sync_probe_wait_kernel(unsigned *ctr, unsigned *result) {
  const unsigned long long t0 = wall_clock64();
  unsigned r = 2;
  while (wall_clock64() - t0 < 200000ull) {
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { r = 1; break; }
    __builtin_amdgcn_s_sleep(8);
  }
  *result = r;
}
static __global__ void fit_sync_probe_post_kernel(unsigned *ctr) { __hip_atomic_store(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// scratch: two device words (zeroed here); returns a HIP error code, *ok = 1 / 0
static inline hipError_t fit_sync_probe(hipStream_t sA, hipStream_t sB, unsigned *scratch, int *ok) {
  hipError_t e;
  if ((e = hipMemsetAsync(scratch, 0, 2 * sizeof(unsigned), sA)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(sA)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(sB)) != hipSuccess) return e;
  hipLaunchKernelGGL(fit_sync_probe_wait_kernel, dim3(1), dim3(1), 0, sB, scratch, scratch + 1);
  hipStreamQuery(sB);                                   // (flush the submission)
  const unsigned long long spin0 = (unsigned long long)clock();
  while ((unsigned long long)clock() - spin0 < (unsigned long long)(CLOCKS_PER_SEC / 5000)) {}      // ~200 us: the waiter is running by now
  hipLaunchKernelGGL(fit_sync_probe_post_kernel, dim3(1), dim3(1), 0, sA, scratch);
  if ((e = hipStreamSynchronize(sA)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(sB)) != hipSuccess) return e;
  unsigned r[2] = {0, 0};
  if ((e = hipMemcpy(r, scratch, sizeof(r), hipMemcpyDeviceToHost)) != hipSuccess) return e;
  *ok = r[1] == 1 ? 1 : 0;
  return hipSuccess;
}
