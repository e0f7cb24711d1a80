// bnx_host.h -- what bnf_api.hip needs of the split-precision translation unit (bnx_api.hip, bnx_kernels.h): the kernel
// instantiations as function pointers, the re-encoding of the two blobs and the perturbation launch in the split layout.
#pragma once
#include <hip/hip_runtime.h>

#include "bnf_build.h"

using BnxMhFn = void (*)(BnfMhArgs);
using BnxEffFn = void (*)(BnfEffArgs);

// mode 0: log posterior, 1: one Metropolis-Hastings iteration; R row tiles per wave and W waves per workgroup of the variant returned
BnxMhFn bnx_mh_fn(int KS, int mode, int *R, int *W);
BnxEffFn bnx_eff_fn(int KSF, int *R, int *W);
// blobx / eblobx <- blob / eblob (allocated on first use); no-op while they are current
int bnx_prepare(BnfState *st, hipStream_t stream);
// bnf_noise_kernel's launch writing hi / lo fp16 at the positions of `posx`
void bnx_launch_noise(const BnfNoiseArgs &na, const int *posx, int n_sets, hipStream_t stream);
