// gvx_comm.h — how the W workers of a training job hand device buffers to each other (internal to libgvk.so).
//
// The engine (gvx_engine.cpp) needs two exchanges and nothing else:
//   * all_gather: after a schedule step every worker has trained a different head partition of one head group, each in
//     ITS slot of the group's slab; one in-place all-gather of the slab gives every worker the whole, current group again
//     (the reference moves the same data GPU -> host scatter -> host gather -> GPU, include/core/solver.h:1349-1428);
//   * all_to_all: random-walk pools — every worker samples a 1/W slice of every block pool, the slices of a block go to
//     the worker that trains it (graph.cuh:298-450 fills all blocks from one host sampler pool instead).
// Three carriers behind one interface:
//   RCCL     over xGMI — one communicator per GPU: ncclCommInitAll when one process drives all GPUs (device_ids = [0, 1,
//            ...]), ncclCommInitRank with a broadcast unique id when there is one process per GPU.  librccl is opened with
//            dlopen at the first multi-GPU build, so a single-GPU process never needs it;
//   copies   one process whose workers share GPUs (device_ids = [0, 0]: the way the multi-worker logic is exercised on a
//            one-GPU box) — device copies ordered by events;
//   callbacks a transport the embedding program supplies (gvx.h gvx_transport): how the CPU tests run the same engine code
//            over gloo.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "gvx.h"

namespace gvx {

struct Peer {  // one LOCAL worker as the comm layer sees it
    int rank;            // its rank among the W workers of the job
    int device;          // HIP device
    hipStream_t stream;  // the stream its exchanges are enqueued on
};

class Comm {
public:
    virtual ~Comm() {}
    // In-place all-gather for all local workers at once: slabs[i] is local worker i's buffer of world * bytes bytes, its
    // own part already at slabs[i] + rank_i * bytes.  Asynchronous: when everything enqueued on local[i].stream so far has
    // completed, slabs[i] is complete.  The caller has made local[i].stream wait for whatever produced its part.
    virtual int all_gather(const std::vector<Peer> &local, const std::vector<void *> &slabs, size_t bytes) = 0;
    // send[i] / recv[i]: world parts of `bytes` bytes each; part q of send[i] arrives as part rank_i of rank q's recv.
    virtual int all_to_all(const std::vector<Peer> &local, const std::vector<const void *> &send,
                           const std::vector<void *> &recv, size_t bytes) = 0;
    virtual const char *name() const = 0;
};

// nullptr + *why when RCCL cannot be used (library missing, a device listed twice, communicator creation failed)
Comm *make_rccl_in_process(const std::vector<int> &devices, std::string *why);
Comm *make_rccl_rank(int rank, int world, int device, const void *unique_id, size_t unique_id_bytes, std::string *why);
Comm *make_copies(int world);
Comm *make_callbacks(const gvx_transport &transport, int rank, int world);
int rccl_unique_id(void *out, size_t capacity, std::string *why);

}  // namespace gvx
