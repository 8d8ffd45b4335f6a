// lcty_comm.hip — the one exchange step of the path when a locus' reads are sharded over GPUs (SURVEY.md §8e level 2):
// run_filter's score of a genotype is a sum over reads (src/solvers/solve.rs:105-119), so every rank prefilters its shard and the
// G-long f64 score vectors are SUM-all-reduced — RCCL over xGMI, on the device buffers, from C++. One process per GPU; the host
// launcher (torch.distributed / MPI / anything) only has to hand rank 0's 128-byte id to the other ranks.
#include <rccl/rccl.h>

#include <memory>

#include "lcty_objects.hpp"

using namespace lcty;

static_assert(sizeof(ncclUniqueId) == LCTY_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");

#define LCTY_NCCL(expr)                                                                                   \
    do {                                                                                                  \
        ncclResult_t r__ = (expr);                                                                        \
        if (r__ != ncclSuccess) ::lcty::fail(LCTY_ERR_RUNTIME, "RCCL error %s at %s:%d: %s", ncclGetErrorString(r__), __FILE__, __LINE__, #expr); \
    } while (0)

struct lcty_comm {
    lcty_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int32_t n_ranks = 0, rank = 0;
};

extern "C" {

int32_t lcty_comm_unique_id(uint8_t* id) {
    return guarded([&] {
        if (!id) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        ncclUniqueId u;
        LCTY_NCCL(ncclGetUniqueId(&u));
        memcpy(id, &u, sizeof(u));
    });
}

int32_t lcty_comm_create(lcty_ctx* ctx, int32_t n_ranks, int32_t rank, const uint8_t* id, lcty_comm** out) {
    return guarded([&] {
        if (!ctx || !id || !out) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (n_ranks < 1 || rank < 0 || rank >= n_ranks) fail(LCTY_ERR_INVALID_INPUT, "rank %d of %d", rank, n_ranks);
        ctx->activate();
        auto c = std::make_unique<lcty_comm>();
        c->ctx = ctx; c->n_ranks = n_ranks; c->rank = rank;
        ncclUniqueId u;
        memcpy(&u, id, sizeof(u));
        LCTY_NCCL(ncclCommInitRank(&c->comm, n_ranks, u, rank));
        *out = c.release();
    });
}

void lcty_comm_destroy(lcty_comm* c) {
    if (!c) return;
    if (c->comm) (void)ncclCommDestroy(c->comm);
    delete c;
}

int32_t lcty_prefilter_allreduce(lcty_reads* reads, lcty_comm* comm) {
    return guarded([&] {
        if (!reads || !comm) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (reads->ctx != comm->ctx) fail(LCTY_ERR_INVALID_INPUT, "the batch and the communicator belong to different contexts");
        if (reads->n_scores == 0) fail(LCTY_ERR_INVALID_INPUT, "lcty_prefilter_async has not been called on this batch");
        reads->ctx->activate();
        reads->check_device_error();
        hipStream_t s = reads->ctx->stream;
        // in place, on the stream the prefilter kernels ran on: ordered behind them without a host synchronisation
        LCTY_NCCL(ncclAllReduce(reads->d_scores.p, reads->d_scores.p, reads->n_scores, ncclDouble, ncclSum, comm->comm, s));
        LCTY_HIP(hipStreamSynchronize(s));
    });
}

}  // extern "C"
