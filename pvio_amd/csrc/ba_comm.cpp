// ba_comm.cpp -- RCCL communicator for landmark-sharded solves (one process per GPU, xGMI).
// The only collective on the data path: all-reduce of [reduced pose system | vectors | scalars] once per
// linearization and of 8 scalars once per back-substitution.  Payloads are <= 1.6 MB,
// i.e. latency-bound on xGMI, so they go out as ONE buffer each.
#include <rccl/rccl.h>

#include <cstring>

#include "ba_solver.h"

namespace pvba {

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

int comm_unique_id(uint8_t id[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return 1;
    std::memcpy(id, &u, 128);
    return 0;
}
int comm_init(Comm **out, const uint8_t id[128], int rank, int world, int device) {
    if (hipSetDevice(device) != hipSuccess) return 1;
    ncclUniqueId u;
    std::memcpy(&u, id, 128);
    Comm *c = new Comm();
    c->rank = rank, c->world = world;
    if (ncclCommInitRank(&c->comm, world, u, rank) != ncclSuccess) {
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}
int comm_allreduce(Comm *c, double *buf, size_t n, int op_max, hipStream_t st) {
    if (!c || !c->comm) return 1;
    return ncclAllReduce(buf, buf, n, ncclDouble, op_max ? ncclMax : ncclSum, c->comm, st) == ncclSuccess ? 0 : 1;
}
void comm_destroy(Comm *c) {
    if (!c) return;
    if (c->comm) ncclCommDestroy(c->comm);
    delete c;
}

} // namespace pvba
