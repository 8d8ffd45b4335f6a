// gather_probe — what a random 32-byte gather costs on this device as a function of the footprint it is spread over, the number of
// wavefronts doing it and the number of gathers a lane keeps in flight. The greedy loop of the solver is one such gather per
// candidate out of a 148 GB record array; this tells latency (a lone wavefront, one gather in flight) from throughput limits.
//   usage: gather_probe <footprint GB> <wavefronts> <active lanes 1..64> <in flight 1|2|4|8> <rounds> [chunked MB]
// "chunked": every wavefront draws from its own window of that many MB (the per-chain record array of the solver: 32 MB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

template <int ILP>
__global__ __launch_bounds__(64) void probe(const uint4* __restrict__ buf, uint64_t n_cells, uint64_t window_cells, uint32_t active, uint32_t rounds,
                                            uint64_t* __restrict__ sink) {
    const uint32_t lane = threadIdx.x;
    if (lane >= active) return;
    uint64_t state = mix(blockIdx.x * 64ull + lane + 1);
    const uint64_t base = window_cells ? (mix(blockIdx.x + 77) % (n_cells / window_cells)) * window_cells : 0;
    const uint64_t span = window_cells ? window_cells : n_cells;
    uint64_t acc = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        uint4 a[ILP], b[ILP];
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            state = mix(state + 0x9e3779b97f4a7c15ull + acc * (ILP == 1));          // ILP 1: the next address depends on the data (a dependent chain)
            const uint64_t cell = base + __umul64hi(state, span);
            a[i] = buf[2 * cell]; b[i] = buf[2 * cell + 1];
        }
#pragma unroll
        for (int i = 0; i < ILP; i++) acc += a[i].x ^ b[i].w;
    }
    sink[blockIdx.x * 64ull + lane] = acc;
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 1.0;
    const uint32_t waves = argc > 2 ? atoi(argv[2]) : 64, active = argc > 3 ? atoi(argv[3]) : 40, ilp = argc > 4 ? atoi(argv[4]) : 1;
    const uint32_t rounds = argc > 5 ? atoi(argv[5]) : 20000;
    const double chunk_mb = argc > 6 ? atof(argv[6]) : 0.0;
    const uint64_t n_cells = static_cast<uint64_t>(gb * (1ull << 30)) / 32;
    const uint64_t window = static_cast<uint64_t>(chunk_mb * (1 << 20)) / 32;
    uint4* buf; uint64_t* sink;
    CK(hipMalloc(&buf, n_cells * 32)); CK(hipMalloc(&sink, waves * 64ull * 8));
    CK(hipMemset(buf, 1, n_cells * 32));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        switch (ilp) {
            case 1: hipLaunchKernelGGL(probe<1>, dim3(waves), dim3(64), 0, 0, buf, n_cells, window, active, rounds, sink); break;
            case 2: hipLaunchKernelGGL(probe<2>, dim3(waves), dim3(64), 0, 0, buf, n_cells, window, active, rounds, sink); break;
            case 4: hipLaunchKernelGGL(probe<4>, dim3(waves), dim3(64), 0, 0, buf, n_cells, window, active, rounds, sink); break;
            default: hipLaunchKernelGGL(probe<8>, dim3(waves), dim3(64), 0, 0, buf, n_cells, window, active, rounds, sink); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double gathers = static_cast<double>(waves) * active * rounds * ilp;
    printf("footprint %7.1f GB window %5.0f MB waves %5u lanes %2u in-flight %u: %8.2f ms  %7.1f ns per round  %7.2f G gathers/s  %7.1f GB/s of 32-B cells\n",
           gb, chunk_mb, waves, active, ilp, best, best * 1e6 / rounds, gathers / best / 1e6, gathers * 32 / best / 1e6);
    return 0;
}
