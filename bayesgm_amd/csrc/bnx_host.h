// bnx_host.h -- what bnf_api.hip needs of the split-precision translation unit (bnx_api.hip, bnx_kernels.h): the kernel
// instantiations as function pointers, the re-encoding of the two blobs and the perturbation launch in the split layout.
#pragma once
#include <hip/hip_runtime.h>

#include "bnf_build.h"

using BnxMhFn = void (*)(BnfMhArgs);
using BnxEffFn = void (*)(BnfEffArgs);

// mode 0: log posterior, otherwise the two log posteriors of a Metropolis-Hastings iteration (MODE 3); R row tiles per wave and W waves per workgroup of the variant returned
BnxMhFn bnx_mh_fn(int KS, int mode, int *R, int *W);
BnxEffFn bnx_eff_fn(int KSF, int *R, int *W);
// mode 3: both log posteriors of a Metropolis-Hastings iteration as independent (item, state) units (bnf_mh_kernel MODE 3)
// the row-wise kernels around it (a: BnxMhStepArgs of bnx_kernels.h, passed by pointer to keep that header out of bnf_api.hip)
struct BnxMhStep {
  float *z, *zprop; const float *lp; long long n, row_base; int q, bs, it, init; float q_sd; const float *q_sd_blocks; uint32_t k0, k1;
  unsigned *acc_count, *acc_blocks;
};
void bnx_launch_propose(const BnxMhStep &a, hipStream_t stream);
void bnx_launch_accept(const BnxMhStep &a, hipStream_t stream);
// blobx / eblobx <- blob / eblob (allocated on first use); no-op while they are current
int bnx_prepare(BnfState *st, hipStream_t stream);
// bnf_noise_kernel's launch writing hi / lo fp16 at the positions of `posx`
void bnx_launch_noise(const BnfNoiseArgs &na, const int *posx, int n_sets, hipStream_t stream);
