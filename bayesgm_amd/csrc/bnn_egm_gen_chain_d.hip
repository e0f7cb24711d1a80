#define BNN_GEN_CHAIN_PART 3
#include "bnn_egm_gen_chain.inc"
