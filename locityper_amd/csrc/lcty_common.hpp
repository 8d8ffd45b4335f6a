// lcty_common.hpp — shared host-side plumbing of liblocityper_hip.so:
// error reporting (mirrors src/err.rs:11-30 categories), HIP checks, device buffers, context.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/locityper_hip.h"

namespace lcty {

#ifdef LCTY_DIAG
constexpr bool kDiag = true;
#else
constexpr bool kDiag = false;
#endif

// Thread-local message of the last failure (lcty_last_error()).
void set_last_error(const std::string& msg);

struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(int32_t code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

#define LCTY_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            ::lcty::fail(LCTY_ERR_RUNTIME, "HIP error %s at %s:%d: %s", hipGetErrorName(e__),   \
                         __FILE__, __LINE__, #expr);                                            \
    } while (0)

// Wraps a C-ABI body: exceptions -> status code + last-error string.
template <typename F>
int32_t guarded(F&& body) {
    try {
        body();
        return LCTY_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("out of host memory");
        return LCTY_ERR_RUNTIME;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return LCTY_ERR_RUNTIME;
    }
}

// Owning device allocation.
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
    }
    void alloc(size_t count) {
        release();
        n = count;
        if (count) LCTY_HIP(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
    }
    void ensure(size_t count) {                      // grow-only: a workspace that survives the call
        if (n < count) alloc(count);
    }
    void ensure_slack(size_t count) {                // the same with a quarter of head-room: what comes next is about as large, seldom larger
        if (n < count) alloc(count + count / 4 + 64);
    }
    void upload(const T* host, size_t count, hipStream_t s, size_t dst_off = 0) {
        if (dst_off + count > n) fail(LCTY_ERR_RUNTIME, "device buffer overflow (%zu + %zu > %zu)", dst_off, count, n);
        if (count) LCTY_HIP(hipMemcpyAsync(p + dst_off, host, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void download(T* host, size_t count, hipStream_t s, size_t src_off = 0) const {
        if (src_off + count > n) fail(LCTY_ERR_RUNTIME, "device buffer overread");
        if (count) LCTY_HIP(hipMemcpyAsync(host, p + src_off, count * sizeof(T), hipMemcpyDeviceToHost, s));
    }
    void zero(hipStream_t s) {
        if (n) LCTY_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
    }
};

// One non-trivial read of one solver chain / a location beyond its second (lcty_solve_kernels.hip)
struct __attribute__((aligned(32))) ChainRec {
    uint32_t rp_cur;                // good-read index (24 bit) | current location << 24 (the only field a move changes)
    uint32_t meta;                  // number of locations (8 bit) | index of location 2 in the chain's ExtraLoc run << 8
    double lp0, lp1;                // ln-probability of locations 0 and 1
    uint32_t win0, win1;            // their windows: w1 | w2 << 16
};
static_assert(sizeof(ChainRec) == 32, "ChainRec layout");
struct __attribute__((aligned(16))) ExtraLoc { double lp; uint32_t win; uint32_t _pad; };
static_assert(sizeof(ExtraLoc) == 16, "ExtraLoc layout");

struct KernelTimer {
    uint64_t launches = 0;
    double total_ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

// The stream a context's calls are issued on: its main stream, unless the calling host thread has been bound to another one
// (the tail thread of lcty_solve_queue works on the side stream: whatever it calls — the error flag of the batch, a wider depth
// table — must neither be issued on nor wait for the main stream, where the next locus is running).
extern thread_local hipStream_t tl_stream;
struct StreamRef {
    hipStream_t main = nullptr;
    operator hipStream_t() const { return tl_stream ? tl_stream : main; }
};
struct StreamScope {                                    // binds the calling thread to a stream for the lifetime of the object (scopes nest)
    hipStream_t before;
    explicit StreamScope(hipStream_t s) : before(tl_stream) { tl_stream = s; }
    ~StreamScope() { tl_stream = before; }
    StreamScope(const StreamScope&) = delete; StreamScope& operator=(const StreamScope&) = delete;
};

}  // namespace lcty

// One GPU. The work of a context is issued on `stream` (see lcty::StreamRef); the queue of loci adds the streams further down.
struct lcty_ctx {
    int device = 0;
    lcty::StreamRef stream;
    hipDeviceProp_t props{};
    lcty::KernelTimer timers[LCTY_K_COUNT];
    std::vector<hipEvent_t> event_pool;
    std::mutex timing_mutex;
    bool timing_on = false;                           // lcty_timing_reset switches the event timing on; a run that never asks pays nothing
    std::map<std::string, int64_t> knobs;             // lcty_ctx_set_knob: limits that tests lower to exercise retry / batching paths
    std::string exact_dump_path;                      // lcty_ctx_set_path "exact_dump": where the exact solver leaves the model of a stage's first chain ("" = nowhere)

    hipEvent_t get_event();
    void fold_oldest(lcty::KernelTimer& t, size_t count);
    // records start/stop events around fn() on the stream; resolved lazily in lcty_timing_get (or here once 256 pairs wait)
    template <typename F>
    void timed(int kernel, F&& fn, hipStream_t on = nullptr) {
        if (!timing_on) { fn(); return; }
        if (!on) on = stream;
        lcty::KernelTimer& t = timers[kernel];
        hipEvent_t a, b;
        {
            std::lock_guard<std::mutex> g(timing_mutex);          // lcty_solve_queue times launches from two host threads
            if (t.pending.size() >= 256) fold_oldest(t, 128);
            a = get_event(); b = get_event();
        }
        LCTY_HIP(hipEventRecord(a, on));
        fn();
        LCTY_HIP(hipEventRecord(b, on));
        std::lock_guard<std::mutex> g(timing_mutex);
        t.pending.emplace_back(a, b);
    }
    // Second stream of the context: the last solver stage of a locus (a few long serial chains) runs here while the next
    // locus of a queue is scored, prefiltered and greedily solved on `stream` (lcty_solve_queue).
    hipStream_t side = nullptr;
    // Order of residence in a queue of loci (lcty_solve_queue): the annealing chains of locus i (side stream) are launched when the
    // greedy chains of locus i + 1 are on the device — the greedy workgroups take the LDS and registers they need and the few
    // annealing workgroups fill the gaps; the other way round the annealing wavefronts, one per SIMD, keep a second greedy
    // wavefront from every SIMD they sit on and a part of the greedy stage waits for a second round.
    struct LaunchGate {
        std::mutex m; std::condition_variable cv;
        uint64_t epoch = 0;             // greedy launches announced so far (or released by the queue)
        uint64_t target = 0;            // what the running tail waits for (0: nothing)
        hipEvent_t ev = nullptr;        // recorded on the main stream just before the greedy loop is launched
        // ... and the other way round for the two INITIALISATIONS: the one of the last stage of locus i (400 chains, 8 ms) goes first, the one
        // of locus i + 1's greedy chains (5 000, 57 ms, on the critical path) behind it. Side by side the long one took 73 ms; beside the
        // greedy LOOP the short one took 216 ms (the CU's L1 path is full of the loop's gathers).
        uint64_t tails_started = 0;     // last stages handed to the side stream so far
        uint64_t tail_inits = 0;        // ... whose initialisation has been issued (or which have ended)
        hipEvent_t init_ev = nullptr;   // recorded on the side stream behind that initialisation
    } gate;
    hipStream_t side_stream() {
        if (!side) LCTY_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        return side;
    }
    // Third stream: everything of the NEXT locus that comes before its chains — scores, run_filter, the cut, the location table — while the
    // greedy chains of the current locus have the main stream (one wavefront per SIMD and most of the LDS, but idle issue slots and 35 KB).
    hipStream_t fore = nullptr;
    hipStream_t fore_stream() {
        if (!fore) LCTY_HIP(hipStreamCreateWithFlags(&fore, hipStreamNonBlocking));
        return fore;
    }
    // The chunks of a batch travel on a stream of their own (lcty_reads_append*): a host thread can fill the batch of the NEXT
    // locus over PCIe while the kernels of the current one have the other two streams (lcty_solve_queue_fed).
    hipStream_t copy = nullptr;
    std::mutex copy_mutex;
    std::mutex ws_mutex;              // sizing / (re)allocation of the solver workspaces: the two lanes of a queue look at the free memory one at a time
    hipStream_t copy_stream() {
        std::lock_guard<std::mutex> lock(copy_mutex);
        if (!copy) LCTY_HIP(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
        return copy;
    }
    // Per-chain device state of the solver stages (lcty_solve_kernels.hip), one per stream of the context. Grow-only and kept between
    // stages and loci: at 1 M read pairs the records of 5 000 chains are ~150 GB, and allocating / freeing that per stage costs
    // more than the stage. lcty_ctx_trim releases it.
    struct SolveWorkspace {
        lcty::DevBuf<lcty::ChainRec> recs; lcty::DevBuf<lcty::ExtraLoc> extra;
        uint32_t extra_cap = 0, extra_for_ploidy = 0;
        lcty::DevBuf<uint16_t> gt; lcty::DevBuf<uint8_t> cgc; lcty::DevBuf<uint32_t> cdepth, cnnt, cseg, ctotw, ovf, cuc; lcty::DevBuf<uint64_t> seeds;
        lcty::DevBuf<double> pri, liks, parts, cww, caln, dbg;
        lcty::DevBuf<uint8_t> init_plan;      // the groups of a diploid stage's initialisation and their chains (lcty_solve_device.hpp: InitGroup, InitChainP)
        void release_all() {
            recs.release(); extra.release(); gt.release(); cgc.release(); cdepth.release(); cuc.release(); cnnt.release(); cseg.release(); ctotw.release(); ovf.release();
            seeds.release(); pri.release(); liks.release(); parts.release(); cww.release(); caln.release(); dbg.release(); init_plan.release(); extra_cap = 0;
        }
    } solve_ws[2];
    // lcty_solve_given (lcty_solve_given.hip): `Solver::solve` is called from the reference's worker threads at once; every call in flight
    // has a slot — a stream, the state of one chain, the caller's arrays on the device, a depth table of its own. Grow-only, kept between
    // calls; lcty_ctx_trim releases the slots that are free.
    struct GivenSlot {
        hipStream_t stream = nullptr;
        bool busy = false;
        SolveWorkspace ws;
        lcty::DevBuf<double> lut; uint32_t lut_depth = 0; uint64_t lut_of = 0;          // serial of the locus the table was made for, or the id of the caller's tables
        bool lut_is_given = false; uint32_t lut_given_width = 0, lut_given_rows = 0;
        lcty::DevBuf<uint64_t> read_ixs; lcty::DevBuf<double> lp, weight; lcty::DevBuf<uint32_t> win; lcty::DevBuf<uint8_t> gc; lcty::DevBuf<uint16_t> assgn;
    };
    std::vector<std::unique_ptr<GivenSlot>> given_slots;
    std::mutex given_mutex;
    // Lane scratch of alignment recovery (lcty_transfer.hip): tens of GB for long reads, kept between the chunks of a streaming batch
    // (allocating it costs more than the kernel). The solver stages take it back before they size their own workspace; lcty_ctx_trim
    // releases it.
    lcty::DevBuf<uint8_t> transfer_scratch;
    lcty::DevBuf<uint8_t> transfer_recs; lcty::DevBuf<uint32_t> transfer_words;      // arenas of the transferred alignments (records as bytes: lcty_aln_rec is declared later)
    // Device buffers of candidate generation (lcty_map.hip: arenas of CIGAR words and chains, kernel scratch — tens of GB for long reads on
    // many alleles), kept from chunk to chunk of a streaming loop { lcty_reads_map_append; lcty_score_reads } and released with the lane
    // scratch above: before the solver stages size their workspace, and by lcty_ctx_trim.
    std::shared_ptr<void> map_scratch;
    std::mutex map_mutex;             // one lcty_reads_map_append at a time per context: its buffers are the context's
    std::mutex scratch_mutex;
    void release_transfer_scratch() {
        std::lock_guard<std::mutex> g(scratch_mutex);
        map_scratch.reset();
        if (transfer_scratch.n) transfer_scratch.release();
        if (transfer_recs.n) transfer_recs.release();
        if (transfer_words.n) transfer_words.release();
    }
    int64_t knob(const char* name, int64_t dflt) const {
        auto it = knobs.find(name);
        return it == knobs.end() ? dflt : it->second;
    }
    // Switches of the developer build only (make DIAG=1 -> -DLCTY_DIAG, liblocityper_hip_diag.so): traces on stderr, shader-clock stamps
    // inside kernels, alternative kernel forms under measurement. The product library knows none of these names and carries none of
    // their code: the branches below are dead there and the timed kernel instantiations are never referenced.
    int64_t diag_knob(const char* name, int64_t dflt) const { return lcty::kDiag ? knob(name, dflt) : dflt; }
    void activate() const { LCTY_HIP(hipSetDevice(device)); }
};
