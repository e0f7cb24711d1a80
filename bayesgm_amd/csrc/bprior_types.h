// bprior_types.h -- plan of the Bayesian conditional-prior network (bprior_kernels.h), shared with the session state (bnn_state.h)
#pragma once

#define BPRIOR_MAX_LAYERS 4
#define BPRIOR_THREADS 256
#define BPRIOR_LEAK 0.2f
#define BPRIOR_EPS 1e-6f
#define BPRIOR_BN_EPS 1e-3f
#define BPRIOR_SCALE_EPS 1.1920928955078125e-07f
#define BPRIOR_TAG_EPS 8u
#define BPRIOR_TAG_SIGN 9u
#define BPRIOR_NET_ID 4u

struct BPriorNet {
  int n_layers;
  int dims[BPRIOR_MAX_LAYERS + 1];                 // n_segments, hidden..., q + 1
  int gamma_off, beta_off;
  int loc_off[BPRIOR_MAX_LAYERS], rho_off[BPRIOR_MAX_LAYERS], bias_off[BPRIOR_MAX_LAYERS];
  int sin_w[BPRIOR_MAX_LAYERS], sout_w[BPRIOR_MAX_LAYERS], words;      // sign-word layout of a row (oracle/bnn.py sign_layout)
  int n_params, wmax, n_kernel;                    // widest layer side; sum of in x out over the layers
  int norm_mode;                                   // 0 statistics of the batch at hand, 1 fixed mean 0 / variance 1
};

