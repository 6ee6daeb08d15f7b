// comm_stub.cpp -- the emulated test build has no RCCL.  The landmark-sharded solve is still testable on CPU: the test
// registers an all-reduce callback (implemented with torch.distributed / gloo in tests/multi_rank_worker.py) and the
// stub forwards every collective to it.  Test infrastructure only.
#include "ba_solver.h"

extern "C" {
typedef int (*hipemu_allreduce_fn)(double *buf, long n, int op_max);
static hipemu_allreduce_fn g_allreduce = nullptr;
void hipemu_set_allreduce(hipemu_allreduce_fn fn) { g_allreduce = fn; }
}

namespace pvba {
struct Comm {
    int rank, world;
};
int comm_unique_id(uint8_t *id) {
    for (int i = 0; i < 128; ++i) id[i] = (uint8_t)i;
    return 0;
}
int comm_init(Comm **out, const uint8_t *, int rank, int world, int) {
    *out = new Comm{rank, world};
    return 0;
}
int comm_allreduce(Comm *c, double *buf, size_t n, int op_max, hipStream_t) {
    if (!c || !g_allreduce) return 1;
    return g_allreduce(buf, (long)n, op_max);
}
void comm_destroy(Comm *c) { delete c; }
} // namespace pvba
