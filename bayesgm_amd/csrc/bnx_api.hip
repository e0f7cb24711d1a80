// bnx_api.hip -- the split-precision ("f16 x 3") instantiations of the Bayesian-network sampling kernels (bnx_kernels.h inside
// bnf_kernels.h's bnf_mh_kernel / bnf_effects_kernel) and the host pieces that differ from bnf_api.hip: the blobs' re-encoding and
// the perturbation launch.  Entered from bnf_api.hip when the session's precision is 2 (bgm_bnn_set_precision).
#include "bgm_host.h"
#include "bnx_host.h"
#include "bnx_kernels.h"

#ifndef BNX_R
#define BNX_R 2
#define BNX_W 8
#endif
#ifndef BNX_ER
#define BNX_ER 2
#define BNX_EW 8
#endif

namespace {
template <int KS>
BnxMhFn mh_ks(int mode) {
  return mode == 0 ? bnf_mh_kernel<KS, BNX_R, BNX_W, 0, false, false, true> : bnf_mh_kernel<KS, BNX_R, BNX_W, 3, false, false, true>;
}
BnxMhStepArgs step_args(const BnxMhStep &b) {
  BnxMhStepArgs a{};
  a.z = b.z; a.zprop = b.zprop; a.lp = b.lp; a.n = b.n; a.row_base = b.row_base; a.q = b.q; a.bs = b.bs; a.it = b.it; a.init = b.init;
  a.q_sd = b.q_sd; a.q_sd_blocks = b.q_sd_blocks; a.k0 = b.k0; a.k1 = b.k1; a.acc_count = b.acc_count; a.acc_blocks = b.acc_blocks;
  return a;
}
}  // namespace

BnxMhFn bnx_mh_fn(int KS, int mode, int *R, int *W) {
  *R = BNX_R; *W = BNX_W;
  switch (KS) {
    case 3: return mh_ks<3>(mode);
    case 4: return mh_ks<4>(mode);
    case 5: return mh_ks<5>(mode);
    case 6: return mh_ks<6>(mode);
    default: return mh_ks<8>(mode);
  }
}

BnxEffFn bnx_eff_fn(int KSF, int *R, int *W) {
  *R = BNX_ER; *W = BNX_EW;
  switch (KSF) {
    case 1: return bnf_effects_kernel<1, BNX_ER, BNX_EW, false, true>;
    case 2: return bnf_effects_kernel<2, BNX_ER, BNX_EW, false, true>;
    case 3: return bnf_effects_kernel<3, BNX_ER, BNX_EW, false, true>;
    case 4: return bnf_effects_kernel<4, BNX_ER, BNX_EW, false, true>;
    default: return bnf_effects_kernel<8, BNX_ER, BNX_EW, false, true>;
  }
}

int bnx_prepare(BnfState *st, hipStream_t stream) {
  if (st->x3_valid) return BGM_OK;
  const BnfPlan &P = st->P;
  if (!st->blobx_dev) BGM_HIP_CHECK(hipMalloc((void **)&st->blobx_dev, sizeof(float) * P.blob_floats));
  if (!st->eblobx_dev) BGM_HIP_CHECK(hipMalloc((void **)&st->eblobx_dev, sizeof(float) * P.e_blob_floats));
  BGM_HIP_CHECK(hipMemsetAsync(st->blobx_dev, 0, sizeof(float) * P.blob_floats, stream));
  BGM_HIP_CHECK(hipMemsetAsync(st->eblobx_dev, 0, sizeof(float) * P.e_blob_floats, stream));
  BnxPackArgs pa{};
  pa.blob = st->blob_dev; pa.blobx = st->blobx_dev; pa.w = st->w_dev; pa.n_w = st->n_w; pa.posx = st->posx_dev;
  pa.frag_floats = P.bias_off; pa.blob_floats = P.blob_floats;
  hipLaunchKernelGGL(bnx_pack_kernel, dim3(64), dim3(256), 0, stream, pa);
  pa.blob = st->eblob_dev; pa.blobx = st->eblobx_dev; pa.w = st->we_dev; pa.n_w = st->n_we; pa.posx = st->posx_e_dev;
  pa.frag_floats = P.e_bias_off; pa.blob_floats = P.e_blob_floats;
  hipLaunchKernelGGL(bnx_pack_kernel, dim3(16), dim3(256), 0, stream, pa);
  BGM_HIP_CHECK(hipGetLastError());
  st->x3_valid = true;
  return BGM_OK;
}

void bnx_launch_noise(const BnfNoiseArgs &na, const int *posx, int n_sets, hipStream_t stream) {
  BnxNoiseArgs b{};
  b.n = na; b.posx = posx;
  hipLaunchKernelGGL(bnx_noise_kernel, dim3((na.n_calls + 1023) / 1024, n_sets), dim3(256), 0, stream, b);
}

void bnx_launch_propose(const BnxMhStep &b, hipStream_t stream) {
  const long long threads = b.n * (4 * ((b.q + 15) >> 4));
  hipLaunchKernelGGL(bnx_propose_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, step_args(b));
}
void bnx_launch_accept(const BnxMhStep &b, hipStream_t stream) {
  hipLaunchKernelGGL(bnx_accept_kernel, dim3((unsigned)((b.n + 255) / 256)), dim3(256), 0, stream, step_args(b));
}
