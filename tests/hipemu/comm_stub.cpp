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
// Inside a stream capture the collective becomes a NODE of the graph, like ncclAllReduce does under hipStreamBeginCapture (RCCL supports
// capture): it runs at every replay, between the kernels recorded around it -- so that graph-captured collectives execute with more than one
// rank somewhere (tests/test_multi_rank_cpu.py::test_sharded_graph_replay_with_captured_collectives).
int comm_allreduce(Comm *c, double *buf, size_t n, int op_max, hipStream_t st) {
    if (!c || !g_allreduce) return 1;
    if (st && st->capturing) {
        hipemu_allreduce_fn fn = g_allreduce;
        st->graph->push_back([fn, buf, n, op_max] { (void)fn(buf, (long)n, op_max); });
        return 0;
    }
    return g_allreduce(buf, (long)n, op_max);
}
void comm_destroy(Comm *c) { delete c; }
} // namespace pvba
