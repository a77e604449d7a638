// comm.h -- the one exchange step of the sharded run, bound to RCCL directly.
//
// The reference has no collective at all: its "distribution" is a task queue of tiles over a
// multiprocessing.managers TCP channel that ships whole pickled tile results to the main process
// (tiling.py:1799-1912).  Here tiles are sharded over the GPUs of a node, and what
// crosses GPUs is small: the recoded overlap strips at shard boundaries (ncclSend / ncclRecv of
// <= 25 MB device buffers, point to point over xGMI), the running maxSegId, the k-means sample /
// centres and the segment histogram (broadcast / all-gather / all-reduce).  Every call works on
// device memory and returns when the operation has completed on the context's stream.
#pragma once
#include "common.h"
#include <rccl/rccl.h>

struct shp_comm {
    shp_ctx *ctx = nullptr;
    ncclComm_t nc = nullptr;
    int rank = 0, world = 1;
};

#define NCCLCHK(ctx, call)                                                                  \
    do {                                                                                    \
        ncclResult_t _r = (call);                                                           \
        if (_r != ncclSuccess)                                                              \
            SHP_FAIL(ctx, SHP_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call,        \
                     ncclGetErrorString(_r));                                               \
    } while (0)

static int comm_finish(shp_comm *cm)
{
    HIPCHK(cm->ctx, hipStreamSynchronize(cm->ctx->stream));
    return 0;
}
