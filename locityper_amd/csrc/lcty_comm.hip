// lcty_comm.hip — the exchange steps of the path over several GPUs. Level 2, the one exchange step of the path when a locus' reads are sharded over GPUs (SURVEY.md §8e level 2):
// run_filter's score of a genotype is a sum over reads (src/solvers/solve.rs:105-119), so every rank prefilters its shard and the
// G-long f64 score vectors are SUM-all-reduced — RCCL over xGMI, on the device buffers, from C++. One process per GPU; the host
// launcher (torch.distributed / MPI / anything) only has to hand rank 0's 128-byte id to the other ranks.
#include <rccl/rccl.h>

#include <algorithm>
#include <limits>
#include <memory>
#include <vector>

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
    DevBuf<int64_t> d_status;          // {status, size, ~size}: what every rank agrees on before a data collective (allocated with the communicator)
};

namespace {

// A rank whose local part failed (a bad read shard, an arena overflow, an allocation, a launch error, a chain that lost its
// hand-shake ...) must not leave the others inside a data collective for ever: every rank joins this three-word MAX all-reduce
// unconditionally, with its own status, and all of them raise when any of them failed. `local` runs the rank's part and may throw.
// EVERYTHING of a rank that can fail between two data collectives belongs in such a lambda. `same` is a number that must be equal
// on all ranks (a count the next collective is sized with): the words are {status, same, ~same}, so max(same) == ~max(~same) iff
// every rank brought the same value — a mismatch would otherwise corrupt memory or hang inside RCCL.
// Test hook: the knob "comm_fail_at" = k makes the k-th agreement (1-based) of the current library call fail on this rank.
struct AgreeScope {
    lcty_comm* comm; int64_t fail_at; int64_t step = 0;
    explicit AgreeScope(lcty_comm* c) : comm(c), fail_at(c->ctx->knob("comm_fail_at", 0)) {}
    template <typename F>
    void then(F&& local, uint64_t same = 0) {
        int32_t rc = LCTY_OK;
        std::string msg;
        step++;
        try {
            if (fail_at == step) fail(LCTY_ERR_RUNTIME, "rank %d: injected failure at agreement %lld (knob comm_fail_at)", comm->rank, static_cast<long long>(step));
            local();
        }
        catch (const Error& e) { rc = e.code; msg = e.what(); }
        catch (const std::exception& e) { rc = LCTY_ERR_RUNTIME; msg = e.what(); }
        comm->ctx->activate();
        hipStream_t s = comm->ctx->stream;
        // the status words were allocated with the communicator: nothing here can fail for a reason local to this rank but the
        // runtime itself
        const int64_t v = static_cast<int64_t>(same & 0x3FFFFFFFFFFFFFFFull);
        int64_t words[3] = {rc, v, ~v}, agreed[3] = {rc, v, ~v};
        comm->d_status.upload(words, 3, s);
        LCTY_NCCL(ncclAllReduce(comm->d_status.p, comm->d_status.p, 3, ncclInt64, ncclMax, comm->comm, s));
        comm->d_status.download(agreed, 3, s);
        LCTY_HIP(hipStreamSynchronize(s));
        if (rc != LCTY_OK) fail(rc, "%s", msg.c_str());
        if (agreed[0] != LCTY_OK) fail(static_cast<int32_t>(agreed[0]), "rank %d: another rank of the communicator failed with status %d before the exchange", comm->rank, static_cast<int>(agreed[0]));
        if (agreed[1] != ~agreed[2]) fail(LCTY_ERR_INVALID_INPUT, "rank %d: the ranks of the communicator disagree about a size of the exchange (%lld here)", comm->rank, static_cast<long long>(v));
    }
};

// the sizes of a stage as one word the ranks must agree on: n_gt (< 2^31 chains per stage) in the high part, attempts below — no two
// (n_gt, attempts) pairs share a word
inline uint64_t stage_word(uint64_t n_gt, uint32_t attempts) { return ((n_gt & 0x3FFFFFFFull) << 32) | attempts; }
// a 62-bit mix of several sizes (what AgreeScope::then compares): ranks that differ in any of them differ in the word
inline uint64_t sizes_word(std::initializer_list<uint64_t> v) {
    uint64_t h = 0x9e3779b97f4a7c15ull;
    for (uint64_t x : v) { h ^= x + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 31; }
    return h & 0x3FFFFFFFFFFFFFFFull;
}

// per-chain likelihoods of the blocks of all ranks, gathered on the devices; everything fallible under agreements
void gather_chain_likelihoods(AgreeScope& A, lcty_comm* comm, const std::vector<double>& local, uint64_t per, uint32_t attempts,
                              uint64_t n_gt, double* lik_mean, double* lik_var, double* liks_out) {
    const uint64_t n_ranks = static_cast<uint64_t>(comm->n_ranks);
    hipStream_t s = comm->ctx->stream;
    DevBuf<double> d_send, d_recv;
    const uint64_t cnt = per * attempts;
    A.then([&] {
        comm->ctx->activate();
        d_send.alloc(std::max<uint64_t>(cnt, 1));
        d_recv.alloc(std::max<uint64_t>(cnt, 1) * n_ranks);
        d_send.upload(local.data(), cnt, s);
    }, cnt);
    if (cnt) LCTY_NCCL(ncclAllGather(d_send.p, d_recv.p, cnt, ncclDouble, comm->comm, s));
    std::vector<double> all(std::max<uint64_t>(cnt, 1) * n_ranks);
    d_recv.download(all.data(), cnt * n_ranks, s);
    LCTY_HIP(hipStreamSynchronize(s));
    // rank r's block starts at r * per in the gathered array and at genotype r * per in the list: the same index
    for (uint64_t g = 0; g < n_gt; g++) {
        const double* l = all.data() + g * attempts;
        math::mean_variance_or_nan(l, attempts, &lik_mean[g], &lik_var[g]);
        if (liks_out) memcpy(liks_out + g * attempts, l, sizeof(double) * attempts);
    }
}

}  // namespace

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
        c->d_status.alloc(4);
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

int32_t lcty_comm_ranks(const lcty_comm* comm, int32_t* n_ranks, int32_t* rank) {
    return guarded([&] {
        if (!comm) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        int n = 0, r = 0;
        LCTY_NCCL(ncclCommCount(comm->comm, &n));
        LCTY_NCCL(ncclCommUserRank(comm->comm, &r));
        if (n_ranks) *n_ranks = n;
        if (rank) *rank = r;
    });
}

int32_t lcty_prefilter_allreduce(lcty_reads* reads, lcty_comm* comm) {
    return guarded([&] {
        if (!reads || !comm) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        AgreeScope A(comm);
        // every check that can differ between the ranks sits behind the agreement: a rank whose prefilter never ran, or ran on
        // another number of genotypes, makes ALL ranks return an error instead of leaving the others in the all-reduce
        A.then([&] {
            if (reads->ctx != comm->ctx) fail(LCTY_ERR_INVALID_INPUT, "the batch and the communicator belong to different contexts");
            if (reads->n_scores == 0) fail(LCTY_ERR_INVALID_INPUT, "lcty_prefilter_async has not been called on this batch");
            reads->ctx->activate();
            reads->check_device_error();
        }, reads->n_scores);
        hipStream_t s = reads->ctx->stream;
        // in place, on the stream the prefilter kernels ran on: ordered behind them without a host synchronisation
        LCTY_NCCL(ncclAllReduce(reads->d_scores.p, reads->d_scores.p, reads->n_scores, ncclDouble, ncclSum, comm->comm, s));
        LCTY_HIP(hipStreamSynchronize(s));
    });
}

// SURVEY.md §8e level 3: the (genotype, attempt) chains of a stage are independent units — the reference deals the genotypes of a
// stage to its worker threads (solve.rs:1052-1062); here a contiguous block of the stage's genotype list goes to every rank, each
// rank runs its block on its own GPU (the scored reads are replicated), and the per-chain likelihoods are all-gathered on the
// devices (RCCL over xGMI); mean and variance are then taken by every rank from the same numbers with the same code, so the
// result does not depend on the number of ranks and equals lcty_solve_stage bit for bit (a chain's random stream is its seed).
int32_t lcty_solve_stage_sharded(lcty_reads* reads, lcty_comm* comm, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy,
                                 const double* priors, const lcty_solver* solver, uint32_t attempts, const uint64_t* chain_seeds,
                                 double* lik_mean, double* lik_var, double* liks_out) {
    return guarded([&] {
        if (!reads || !comm || !genotypes || !solver || !chain_seeds || !lik_mean || !lik_var) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (attempts == 0 || ploidy == 0) fail(LCTY_ERR_INVALID_INPUT, "attempts and ploidy must be positive");
        AgreeScope A(comm);
        const uint64_t n_ranks = static_cast<uint64_t>(comm->n_ranks), rank = static_cast<uint64_t>(comm->rank);
        const uint64_t per = (n_gt + n_ranks - 1) / n_ranks;                 // block partition of the stage's genotype list
        const uint64_t lo = std::min(rank * per, n_gt), hi = std::min(lo + per, n_gt);
        std::vector<double> local;
        A.then([&] {
            if (reads->ctx != comm->ctx) fail(LCTY_ERR_INVALID_INPUT, "the batch and the communicator belong to different contexts");
            local.assign(std::max<uint64_t>(per * attempts, 1), std::numeric_limits<double>::quiet_NaN());
            if (hi <= lo) return;
            std::vector<double> m(hi - lo), v(hi - lo);
            const int32_t rc = lcty_solve_stage(reads, genotypes + lo * ploidy, hi - lo, ploidy, priors ? priors + lo : nullptr, solver,
                                                attempts, chain_seeds + lo * attempts, m.data(), v.data(), local.data());
            if (rc != LCTY_OK) fail(rc, "%s", lcty_last_error());
        }, stage_word(n_gt, attempts));
        gather_chain_likelihoods(A, comm, local, per, attempts, n_gt, lik_mean, lik_var, liks_out);
    });
}

// SURVEY.md §8e level 2, the whole path: every rank scored ITS shard of the locus' reads (contiguous blocks of the read list, in
// rank order). The prefilter's exchange is lcty_prefilter_allreduce; the solver stages need, for every allele of the stage's
// genotypes, the possible locations of EVERY good read pair — GenotypeAlignments::new walks all reads of the locus
// (assgn.rs:41-84). Exchange: the rows of those alleles of every rank's location table, packed (32-byte cells + the run of
// further pair-alignments), all-gathered in chunks of rows over RCCL and laid side by side into one table per rank; then the
// stage's chains are dealt to the ranks in contiguous blocks as in lcty_solve_stage_sharded and their likelihoods all-gathered.
// The result equals lcty_solve_stage on the unsharded batch bit for bit (the table cells are the same numbers in the same order).
// Everything a rank does between two collectives — allocations, kernel launches, the plan — runs inside an agreement.
int32_t lcty_solve_stage_read_sharded(lcty_reads* shard, lcty_comm* comm, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy,
                                      const double* priors, const lcty_solver* solver, uint32_t attempts, const uint64_t* chain_seeds,
                                      double* lik_mean, double* lik_var, double* liks_out) {
    return guarded([&] {
        if (!shard || !comm || !genotypes || !solver || !chain_seeds || !lik_mean || !lik_var) fail(LCTY_ERR_INVALID_INPUT, "null argument");
        if (attempts == 0 || ploidy == 0) fail(LCTY_ERR_INVALID_INPUT, "attempts and ploidy must be positive");
        AgreeScope A(comm);
        const uint32_t n_ranks = static_cast<uint32_t>(comm->n_ranks), rank = static_cast<uint32_t>(comm->rank);
        hipStream_t s = comm->ctx->stream;
        std::unique_ptr<RowGatherer> G;
        uint64_t mine[2] = {0, 0};
        DevBuf<uint64_t> d_mine, d_all;
        A.then([&] {
            if (shard->ctx != comm->ctx) fail(LCTY_ERR_INVALID_INPUT, "the batch and the communicator belong to different contexts");
            shard->ctx->activate();
            G = std::make_unique<RowGatherer>(shard, genotypes, n_gt, ploidy);
            G->count(shard, rank, &mine[0], &mine[1]);
            d_mine.alloc(2); d_all.alloc(2ull * n_ranks);
            d_mine.upload(mine, 2, s);
        }, stage_word(n_gt, attempts));
        // sizes of every shard
        LCTY_NCCL(ncclAllGather(d_mine.p, d_all.p, 2, ncclUint64, comm->comm, s));
        std::vector<uint64_t> all(2ull * n_ranks), goods(n_ranks), extras(n_ranks);
        d_all.download(all.data(), all.size(), s);
        LCTY_HIP(hipStreamSynchronize(s));
        for (uint32_t r = 0; r < n_ranks; r++) { goods[r] = all[2 * r]; extras[r] = all[2 * r + 1]; }
        // the plan allocates the staging and the table of this rank; the first chunk is packed under the same agreement, every later
        // chunk together with the placement of the one before it
        A.then([&] {
            G->plan(goods.data(), extras.data(), n_ranks);
            if (G->n_rows) G->pack_chunk(shard, rank, 0, G->send_cells(), shard->gather.send_pa.p);
        });
        // the plan's sizes come from a per-context knob (gather_chunk_mb) and from every rank's own genotype list: ranks that differ in
        // any of them would enter the first all-gather with different sizes or loop a different number of times — agreed BEFORE it
        A.then([] {}, sizes_word({G->n_rows, G->rows_per_chunk, G->chunk_cells(), G->ext_stride}));
        for (uint32_t row0 = 0; row0 < G->n_rows; row0 += G->rows_per_chunk) {
            LCTY_NCCL(ncclAllGather(G->send_cells(), G->recv_cells(0), G->chunk_cells() * 32, ncclUint8, comm->comm, s));
            const uint32_t next = row0 + G->rows_per_chunk;
            A.then([&] {
                for (uint32_t r = 0; r < n_ranks; r++) G->place_chunk(G->recv_cells(r), r, row0);
                if (next < G->n_rows) G->pack_chunk(shard, rank, next, G->send_cells(), shard->gather.send_pa.p);
            }, G->chunk_cells());
        }
        LCTY_NCCL(ncclAllGather(shard->gather.send_pa.p, shard->gather.pa.p, G->ext_stride * sizeof(PairAlnDev), ncclUint8, comm->comm, s));

        const uint64_t per = (n_gt + n_ranks - 1) / n_ranks;                 // block partition of the stage's genotype list
        const uint64_t lo = std::min<uint64_t>(static_cast<uint64_t>(rank) * per, n_gt), hi = std::min(lo + per, n_gt);
        std::vector<double> local;
        A.then([&] {
            G->finish();
            local.assign(std::max<uint64_t>(per * attempts, 1), std::numeric_limits<double>::quiet_NaN());
            if (hi <= lo) return;
            std::vector<double> m(hi - lo), v(hi - lo);
            solve_stage_gathered(shard, *G, genotypes + lo * ploidy, hi - lo, ploidy, priors ? priors + lo : nullptr, solver, attempts,
                                 chain_seeds + lo * attempts, m.data(), v.data(), local.data());
        });
        gather_chain_likelihoods(A, comm, local, per, attempts, n_gt, lik_mean, lik_var, liks_out);
    });
}

}  // extern "C"
