// comm_host.h -- what the translation units with a data-parallel minibatch loop need of comm_api.hip
#pragma once
#include <hip/hip_runtime.h>

// in-place sum over the communicator's ranks, enqueued on `stream`; comm is an ncclComm_t handed through the C ABI as void *
int bgm_comm_enqueue_all_reduce(void *comm, float *buf, long long count, hipStream_t stream);
// ranks of the communicator (and this process's), for argument checks
int bgm_comm_world(void *comm, int *world, int *rank);
