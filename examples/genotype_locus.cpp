// examples/genotype_locus.cpp — the whole path through the C ABI from compiled code, no Python:
// synthetic locus (BASELINE configs[0] shape by default) -> lcty_locus_create -> lcty_reads_create/append ->
// lcty_score_reads -> lcty_solve -> genotype call. This is the call sequence INTEGRATION.md describes for the Rust side.
//
//   g++ -O2 -std=c++17 -Iinclude examples/genotype_locus.cpp -o examples/genotype_locus \
//       -Llocityper_amd -llocityper_hip -Llocityper_amd/synth -llcty_synth -Wl,-rpath,... (see tests/test_gpu_example.py)
//   ./genotype_locus [n_alleles] [n_pairs]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "locityper_hip.h"

extern "C" {   // locityper_amd/synth/lcty_synth.c: the synthetic workloads of SURVEY.md §8(d)
struct synth_locus;
synth_locus* synth_locus_new(uint64_t seed, uint32_t n_alleles, uint32_t base_len, uint32_t k, int technology, uint32_t read_len,
                             uint64_t n_pairs);
void synth_locus_free(synth_locus*);
const uint8_t* synth_locus_seqs(const synth_locus*);
const uint64_t* synth_locus_seq_off(const synth_locus*);
const uint16_t* synth_locus_counts(const synth_locus*);
const uint64_t* synth_locus_cnt_off(const synth_locus*);
const lcty_bg* synth_locus_bg(const synth_locus*);
void synth_locus_true_genotype(const synth_locus*, uint32_t* out);
void synth_reads_sizes(const synth_locus*, uint64_t first, uint64_t n, uint32_t* mate_len, uint32_t* rec_cnt, uint32_t* cig_cnt);
void synth_reads_fill(const synth_locus*, uint64_t first, uint64_t n, const uint64_t* mate_off, uint32_t* bases2, uint32_t* nmask,
                      const uint64_t* aln_off, lcty_aln_rec* recs, const uint64_t* cigar_off, uint32_t* cigar);
}

static void ok(int32_t rc, const char* what) {
    if (rc != LCTY_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, lcty_last_error()); std::exit(1); }
}

int main(int argc, char** argv) {
    const uint32_t n_alleles = argc > 1 ? static_cast<uint32_t>(std::atoi(argv[1])) : 8;
    const uint64_t n_pairs = argc > 2 ? static_cast<uint64_t>(std::atoll(argv[2])) : 10000;
    const uint32_t k = 25;
    synth_locus* S = synth_locus_new(0x10C17E9E20250001ull, n_alleles, 50000, k, LCTY_TECH_ILLUMINA, 150, n_pairs);
    if (!S) { std::fprintf(stderr, "synthetic locus failed\n"); return 1; }

    // read pairs with all their candidate alignments, as a flat table (what the BAM loop of AllAlignments::load sees)
    std::vector<uint32_t> mate_len(2 * n_pairs), rec_cnt(n_pairs), cig_cnt(n_pairs);
    synth_reads_sizes(S, 0, n_pairs, mate_len.data(), rec_cnt.data(), cig_cnt.data());
    std::vector<uint64_t> mate_off(2 * n_pairs + 1, 0), aln_off(n_pairs + 1, 0), cigar_off(n_pairs + 1, 0);
    for (uint64_t i = 0; i < 2 * n_pairs; i++) mate_off[i + 1] = mate_off[i] + (static_cast<uint64_t>(mate_len[i]) + 31) / 32 * 32;
    for (uint64_t i = 0; i < n_pairs; i++) { aln_off[i + 1] = aln_off[i] + rec_cnt[i]; cigar_off[i + 1] = cigar_off[i] + cig_cnt[i]; }
    std::vector<uint32_t> bases2(mate_off.back() / 16 + 1), nmask(mate_off.back() / 32 + 1), cigar(cigar_off.back() + 1);
    std::vector<lcty_aln_rec> recs(aln_off.back() + 1);
    synth_reads_fill(S, 0, n_pairs, mate_off.data(), bases2.data(), nmask.data(), aln_off.data(), recs.data(), cigar_off.data(), cigar.data());

    lcty_ctx* ctx = nullptr;
    ok(lcty_ctx_create(0, &ctx), "lcty_ctx_create");
    lcty_bg bg = *synth_locus_bg(S);
    lcty_params prm;
    lcty_params_default(&prm);
    ok(lcty_params_resolve(&prm, &bg), "lcty_params_resolve");
    lcty_locus* locus = nullptr;
    ok(lcty_locus_create(ctx, n_alleles, synth_locus_seqs(S), synth_locus_seq_off(S), synth_locus_counts(S), synth_locus_cnt_off(S), k, &bg,
                         &prm, &locus), "lcty_locus_create");
    lcty_reads* reads = nullptr;
    ok(lcty_reads_create(locus, n_pairs, mate_off.back() + 64, aln_off.back() + 1, cigar_off.back() + 1, &reads), "lcty_reads_create");
    lcty_reads_host chunk{};
    chunk.n_pairs = n_pairs;
    chunk.mate_len = mate_len.data(); chunk.mate_off = mate_off.data(); chunk.bases2 = bases2.data(); chunk.nmask = nmask.data();
    chunk.aln_off = aln_off.data(); chunk.recs = recs.data(); chunk.cigar_off = cigar_off.data(); chunk.cigar = cigar.data();
    ok(lcty_reads_append(reads, &chunk), "lcty_reads_append");
    ok(lcty_score_reads(reads), "lcty_score_reads");

    lcty_stage stages[2];
    uint32_t n_stages = 0;
    ok(lcty_stages_default(stages, &n_stages), "lcty_stages_default");
    lcty_call call;
    ok(lcty_solve(reads, 2, stages, n_stages, 1, nullptr, &call, nullptr, nullptr, nullptr), "lcty_solve");

    const uint64_t G = lcty_count_genotypes(n_alleles, 2);
    std::vector<uint16_t> gts(G * 2);
    ok(lcty_generate_genotypes(n_alleles, 2, gts.data(), G), "lcty_generate_genotypes");
    uint32_t truth[2];
    synth_locus_true_genotype(S, truth);
    const uint16_t* g = gts.data() + call.ixs[0] * 2;
    std::printf("called %u,%u true %u,%u quality %.1f ln_prob %.6f good_pairs %llu unexplained %u warnings %u reported %llu\n", g[0], g[1],
                truth[0], truth[1], call.quality, call.ln_probs[0], static_cast<unsigned long long>(call.n_good), call.unexpl_reads,
                call.warnings, static_cast<unsigned long long>(call.n_out));
    const bool right = g[0] == truth[0] && g[1] == truth[1];
    lcty_reads_destroy(reads);
    lcty_locus_destroy(locus);
    lcty_ctx_destroy(ctx);
    synth_locus_free(S);
    return right ? 0 : 2;
}
