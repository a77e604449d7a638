// comm.h -- the one exchange step of the sharded run, bound to RCCL directly.
//
// The reference has no collective at all: its "distribution" is a task queue of tiles over a
// multiprocessing.managers TCP channel that ships whole pickled tile results to the main process
// (tiling.py:1799-1912).  Here tiles are sharded over the GPUs of a node, and what
// crosses GPUs is small: the recoded overlap strips at shard boundaries (ncclSend / ncclRecv of
// <= 25 MB device buffers, point to point over xGMI), the running maxSegId, the k-means sample /
// centres and the segment histogram (broadcast / all-gather / all-reduce).  The collectives and the
// blocking send / recv work on device memory and return when the operation has completed on the
// context's stream.  The strips of the parallel stitch go the asynchronous way instead
// (shp_comm_isend / shp_comm_irecv): the operation is enqueued on the communicator's OWN stream,
// ordered against the chain's stream by events on the device -- a send waits there for the chain
// step that wrote the strip, the chain step that reads a received strip waits for its arrival --
// and the host, i.e. the chain that issues the steps, never waits for a neighbour.
#pragma once
#include "common.h"
#include <rccl/rccl.h>

struct shp_comm {
    shp_ctx *ctx = nullptr;
    ncclComm_t nc = nullptr;
    int rank = 0, world = 1;
    hipStream_t cstream = nullptr;      // the asynchronous operations' stream (created with the communicator)
    std::vector<hipEvent_t> evpool;     // events of the asynchronous operations, reused round robin
    size_t evnext = 0;
};

static hipEvent_t comm_event(shp_comm *cm)
{
    if (cm->evpool.size() < 64) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        cm->evpool.push_back(e);
        return e;
    }
    return cm->evpool[cm->evnext++ % cm->evpool.size()];
}

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
