// lcty_objects.hpp — the opaque handles behind include/locityper_hip.h.
#pragma once

#include <memory>
#include <mutex>
#include <vector>

#include "lcty_common.hpp"
#include "lcty_device.hpp"
#include "lcty_math.hpp"

// ContigSet + KmerCounts + ContigInfos + UniqueKmers + InsertDistr + EditDistCache + DistrCache
struct lcty_locus {
    lcty_ctx* ctx = nullptr;
    uint64_t serial = 0;                     // unique per created locus (a later locus can sit at the address of a destroyed one)
    uint32_t n_alleles = 0, k = 0;
    lcty_bg bg{};
    lcty_params prm{};

    // host copies
    std::vector<uint32_t> allele_len;
    std::vector<uint32_t> ci_off;            // [A+1]
    std::vector<uint32_t> n_windows, reg_start;
    double uniq_mult = 0, compl_mult = 0;
    uint32_t left_padding = 0, half_neighb = 0;
    uint64_t n_unique = 0;
    uint32_t undef_in_set = 0;

    lcty::math::NBinom ins;
    std::vector<double> ins_lut;
    double insert_penalty = 0;

    std::mutex edit_mutex;
    std::vector<uint2> edit_cache;           // (good, passable) per read length; (~0,~0) = not computed
    uint32_t edit_dev_size = 0;

    // device
    lcty::DevBuf<uint32_t> d_allele_len, d_ci_off;
    lcty::DevBuf<uint16_t> d_compl_cnt;
    lcty::DevBuf<uint8_t> d_gc;
    lcty::DevBuf<uint32_t> d_uniq_cnt;
    // the two factors of a window weight as tables over uniq_cnt / compl_cnt (+ a last wk entry of 0: "trivial"): the greedy
    // loop keeps them in LDS. Valid when both counts fit nine bits.
    lcty::DevBuf<double> d_wk, d_wc;
    bool weight_tables_valid = false;
    std::shared_ptr<void> map_index;         // lcty_map.hip: k-mer index of the basis alleles (lcty_locus_build_map_index)
    lcty::DevBuf<uint64_t> d_kset;
    uint64_t kset_cap = 0;
    lcty::DevBuf<double> d_ins_lut;
    lcty::DevBuf<uint2> d_edit_lut;
    lcty::DevBuf<double> d_depth_lut;        // [101][256]
    lcty::DevBuf<lcty::DepthNB> d_depth_nb;  // [101]
    lcty::DevBuf<uint32_t> d_n_windows, d_reg_start;   // [A]
    lcty::DevBuf<double> d_win_weight;       // per position: ContigInfo::neighb_info weight (windows.rs:439-445)
    // ExplicitWeights (windows.rs:196-250) once lcty_locus_set_explicit_weights was called: len + 1 values per allele
    lcty::DevBuf<double> d_ew_val;
    lcty::DevBuf<uint64_t> d_ew_off;         // [A+1]
    bool has_explicit = false;
    // alignment recovery (lcty_transfer.hip): allele sequences stay resident, haplotype-to-haplotype alignments + indices
    lcty::DevBuf<uint8_t> d_seqs;
    lcty::DevBuf<uint64_t> d_seq_off;
    lcty::DevBuf<uint32_t> d_hap_cell_of, d_hap_item_off, d_hap_sparse_off, d_hap_best_off, d_hap_best_ids, d_hap_len;
    lcty::DevBuf<uint4> d_hap_best_meta;     // per (contig, target in best order): {target, first item, items, first sparse entry of the direction}
    lcty::DevBuf<uint2> d_hap_items, d_hap_positions, d_hap_sparse;
    uint32_t hap_transfer_fails = 0, hap_cells = 0;
    bool has_hap_alns = false;
    lcty::DevBuf<double> d_lut_ext;          // [101][lut_ext_depth] depth table of the solver stages (lcty_solve_kernels.hip)
    uint32_t lut_ext_depth = 0;
    uint32_t max_n_windows = 0;

    lcty::LocusView view() const;
    // makes sure (good, passable) is known on the device for every length in `lens`
    void ensure_edit_thresholds(const uint32_t* lens, size_t n);
};

// device-resident batch of read pairs + products of load()
struct lcty_reads {
    lcty_locus* locus = nullptr;
    lcty_ctx* ctx = nullptr;
    uint64_t cap_pairs = 0, cap_bases = 0, cap_recs = 0, cap_cigar = 0;
    uint64_t n_pairs = 0, n_bases = 0, n_recs = 0, n_cigar = 0;
    uint32_t max_recs_per_pair = 0;
    uint32_t max_cigar_per_pair = 0;
    uint32_t max_cigar_per_rec = 0;
    uint64_t recover_dp_cells = 0;                 // aligner cells of the last recovery (lcty_recover_dp_cells)
    uint64_t recover_level_pairs[3] = {0, 0, 0};   // pairs the transfer kernel took at each scratch level (lcty_recover_stats)
    bool scored = false;
    bool counted = false;                    // the records are lcty_aln_counted entries (lcty_reads_append_counted): no CIGAR words
    // streaming batch (lcty_reads_create_streaming): the record / CIGAR / base buffers hold one chunk at a time, the products
    // of every scored chunk stay. Pairs before raw_first have been scored and their raw data dropped; the per-pair raw arrays
    // (mate_len, mate_off, aln_off, cigar_off, pair_meta) are indexed relative to raw_first. raw_first == 0 otherwise.
    bool streaming = false;
    bool pa_pooled = false;                  // the arena has room for chunk-wise reservation by the scoring kernel (lcty_device.hpp: PA_CHUNK)
    uint64_t raw_first = 0, cap_raw_pairs = 0;
    uint64_t chunk_cap_recs = 0, chunk_cap_cigar = 0;   // record / CIGAR capacity of a chunk as given at creation (recovery replaces the tables)
    unsigned long long pa_at_raw_first = 0;   // arena cursor when the current chunk started (a chunk can be scored again)
    // a streaming batch keeps a second set of record tables: alignment recovery merges a chunk's records with the transferred ones INTO
    // the spare set and swaps the two, instead of allocating (and freeing) a chunk's worth of device memory per chunk — 22 GB at
    // 32 768 10-kb reads x 256 alleles, a third of a second a time
    lcty::DevBuf<uint64_t> spare_aln_off, spare_cigar_off;
    lcty::DevBuf<lcty_aln_rec> spare_recs;
    lcty::DevBuf<uint32_t> spare_cigar;
    lcty::DevBuf<uint2> spare_pair_meta;

    lcty::DevBuf<uint32_t> d_mate_len;
    lcty::DevBuf<uint64_t> d_mate_off;
    lcty::DevBuf<uint32_t> d_bases2, d_nmask;
    lcty::DevBuf<uint64_t> d_aln_off;
    lcty::DevBuf<lcty_aln_rec> d_recs;
    lcty::DevBuf<uint64_t> d_cigar_off;
    lcty::DevBuf<uint32_t> d_cigar;
    lcty::DevBuf<uint2> d_pair_meta;
    lcty::DevBuf<uint8_t> d_park;            // scoring kernel, pairs whose saved alignments do not fit the LDS (lcty_score.hip)

    lcty::DevBuf<uint8_t> d_status;
    lcty::DevBuf<double> d_weight, d_unmapped;
    lcty::DevBuf<uint16_t> d_uniq;
    lcty::DevBuf<double> d_matrix;           // [R][A]
    lcty::DevBuf<lcty::PairAlnDev> d_pa;
    lcty::DevBuf<unsigned long long> d_pa_count;
    lcty::DevBuf<uint64_t> d_pa_off;
    lcty::DevBuf<uint32_t> d_pa_cnt;
    lcty::DevBuf<uint32_t> d_pa_idx;         // [R][A]
    // solver stages: compact list of GOOD pairs (AllAlignments::reads order), built lazily after scoring
    lcty::DevBuf<uint32_t> d_good_ix, d_good_cnt;
    uint64_t n_good_cached = 0;
    bool good_valid = false;
    void ensure_good_index();
    // allele-major location table of the solver stages (lcty_solve_kernels.hip), rows of ngp entries
    lcty::DevBuf<uint8_t> d_loc_table;       // [A][ngp] 16-byte LocCell cells
    lcty::DevBuf<uint32_t> d_loc_ext;        // [A][ngp] arena index of a cell's second pair-alignment
    lcty::DevBuf<double> d_loc_unm;          // [ngp] "both mates unmapped" probability of every good read pair
    uint64_t ngp = 0;
    bool loc_table_valid = false;
    uint64_t stat_chains = 0, stat_iterations = 0, stat_accepted = 0;   // last lcty_solve_stage
    lcty::DevBuf<uint16_t> d_unexpl_ids; lcty::DevBuf<unsigned long long> d_unexpl_out;   // count_unexplained_reads scratch
    // the location-table rows of a stage's alleles over the reads of EVERY shard of the locus (lcty_solve_stage_read_sharded /
    // lcty_solve_stage_from_shards): kept on the shard that asked, grow-only
    struct RowGatherBufs {
        lcty::DevBuf<uint8_t> table, send, recv;            // 16-byte cells [rows][ngp]; 32-byte travelling cells: one chunk of rows of this shard; of every shard
        lcty::DevBuf<uint32_t> ext; lcty::DevBuf<double> unm; // the side arrays of the gathered table
        lcty::DevBuf<lcty::PairAlnDev> pa, send_pa;         // [shards][ext_stride]; this shard's run
        lcty::DevBuf<uint16_t> alleles, row_of;
        lcty::DevBuf<unsigned long long> counters;          // [shards] extras counted / handed out
    } gather;
    // run_filter as an integer Gram contraction on the matrix cores (lcty_gram.hip), grow-only
    struct GramBufs {
        lcty::DevBuf<uint32_t> bits, res_list;               // [A][columns / 32]; residual rows in order
        lcty::DevBuf<double> delta, c_part, res_rows, res_scores;
        lcty::DevBuf<uint8_t> dig, residual;                 // [planes][columns]; per row: left to the f64 kernel
        lcty::DevBuf<unsigned long long> counters, S;        // [G] integer sums
        lcty::DevBuf<int> f;
    } gram;
    lcty::DevBuf<uint32_t> d_err;
    lcty::DevBuf<uint32_t> d_defer_list;     // pairs the lean scoring kernel leaves to the general one (lcty_score.hip), grow-only
    lcty::DevBuf<unsigned int> d_defer_count;
    lcty::DevBuf<uint32_t> d_defer_list2;    // pairs the two-slot form of the lean kernel leaves to the general one
    lcty::DevBuf<unsigned int> d_defer_count2;
    lcty::DevBuf<unsigned long long> d_score_dbg;            // lcty_ctx_set_knob "score_timing"
    lcty::DevBuf<double> d_recover_w;        // per pair: read weight when the pair reaches recover_and_group_alignments, else -1

    // prefilter products
    lcty::DevBuf<double> d_scores;           // [G]
    // lcty_prefilter_truncate's sort buffers: grow-only on the batch (an allocation or a release in the head of a queued locus would
    // wait for the annealing chains of the locus before it)
    struct SelectBufs { lcty::DevBuf<uint64_t> k_in, k_out, v_in, v_out; lcty::DevBuf<uint8_t> tmp; lcty::DevBuf<unsigned long long> out; } select;
    lcty::DevBuf<double> d_partials;         // [splits][G]
    uint64_t n_scores = 0;

    lcty::ReadsView view() const;
    void check_device_error(hipStream_t on = nullptr);   // throws when a kernel raised LCTY_ERR_* (looked up on `on`, or the calling thread's stream)
};

namespace lcty {
void launch_score_reads(lcty_reads* reads);
void launch_prefilter_diploid(lcty_reads* reads);                               // all (i <= j) pairs
bool launch_prefilter_gram(lcty_reads* reads);                                  // lcty_gram.hip: the same on the matrix cores, when it applies
void launch_prefilter_tile(lcty_reads* reads, const double* M, uint64_t R, double* d_scores_out);
void launch_prefilter_generic(lcty_reads* reads, const uint16_t* d_genotypes, uint64_t n_gt, uint32_t ploidy,
                              const double* d_priors, double* d_scores);
void compact_matrix(lcty_reads* reads, double* d_out, uint64_t n_good);         // [A][n_good]
uint64_t count_genotypes(uint32_t n_alleles, uint32_t ploidy);
// a chunk whose records, CIGAR words and bases are on the device already (lcty_map.hip -> lcty_reads.hip)
struct DeviceRecords {
    const lcty_aln_rec* recs; const uint32_t* cigar; const uint32_t* bases2; const uint32_t* nmask;      // device
    const uint32_t* n_recs_mate;                     // host, [2 * n_pairs]: records of every read end (its primary or unmapped record first)
    uint32_t max_rec_cigar;
};
int32_t reads_append_device(lcty_reads* R, const lcty_reads_host* h, const DeviceRecords* dev);

// The rows of a stage's alleles of the location tables of several batches (shards of one locus' reads, in read order), laid
// side by side into one table the solver kernels run on (lcty_solve_kernels.hip). The caller moves the packed rows between devices
// (lcty_comm.hip) or hands every shard over itself (lcty_solve_stage_from_shards).
struct RowGatherer {
    lcty_reads* owner; lcty_ctx* ctx; hipStream_t stream;
    std::vector<uint16_t> alleles, row_of;                  // distinct alleles of the stage's genotypes; allele -> row (0xFFFF: none)
    uint32_t n_rows = 0, n_shards = 0, rows_per_chunk = 0;
    std::vector<uint64_t> goods, first;                     // good read pairs of every shard; where a shard's reads start
    uint64_t stride = 0, ext_stride = 0, ngp = 0, n_good = 0;
    RowGatherer(lcty_reads* owner, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy);
    void count(lcty_reads* shard, uint32_t slot, uint64_t* n_good, uint64_t* n_extras);     // builds the shard's table first
    void plan(const uint64_t* goods, const uint64_t* extras, uint32_t n_shards);
    uint8_t* send_cells() const { return owner->gather.send.p; }
    uint8_t* recv_cells(uint32_t shard) const { return owner->gather.recv.p + static_cast<size_t>(shard) * rows_per_chunk * stride * 32; }
    size_t chunk_cells() const { return static_cast<size_t>(rows_per_chunk) * stride; }
    PairAlnDev* run_of(uint32_t shard) const { return owner->gather.pa.p + static_cast<size_t>(shard) * ext_stride; }
    void pack_chunk(lcty_reads* shard, uint32_t slot, uint32_t row0, uint8_t* cells, PairAlnDev* run);   // rows row0.. of one chunk
    void place_chunk(const uint8_t* cells, uint32_t shard, uint32_t row0);
    void finish();
};
void solve_stage_gathered(lcty_reads* owner, const RowGatherer& G, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy,
                          const double* priors, const lcty_solver* solver, uint32_t attempts, const uint64_t* chain_seeds,
                          double* lik_mean, double* lik_var, double* liks_out);
}  // namespace lcty
