/*
 * locityper_hip.h — C ABI of liblocityper_hip.so
 *
 * MI355X (gfx950) implementation of ONE hot path of Locityper: read -> haplotype
 * likelihood scoring and genotype pre-filtering / assignment of `locityper genotype`.
 * Every entry point below names the reference interface it replaces
 * (paths relative to the reference crate root, tprodanov/locityper v1.7.2).
 *
 * Conventions
 *   - all functions return int32 status (LCTY_OK == 0); the message of the last
 *     failure on the calling thread is available from lcty_last_error();
 *     codes mirror the error categories of src/err.rs:11-30;
 *   - handles are opaque and owned by the library; input buffers are caller-owned
 *     host memory, only read during the call; output buffers are caller-allocated;
 *   - distinct handles may be used from distinct threads concurrently;
 *   - nothing in here falls back to the CPU: without a usable HIP device every
 *     compute entry point fails with LCTY_ERR_RUNTIME.
 */
#ifndef LOCITYPER_HIP_H
#define LOCITYPER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (src/err.rs:11-30) ---------------------------------- */
#define LCTY_OK                0
#define LCTY_ERR_INVALID_INPUT 1   /* Error::InvalidInput */
#define LCTY_ERR_INVALID_DATA  2   /* Error::InvalidData  */
#define LCTY_ERR_RUNTIME       3   /* Error::RuntimeError */
#define LCTY_ERR_SOLVER        4   /* Error::Solver       */
#define LCTY_ERR_UNSUPPORTED   5   /* shape outside what this build handles (fails loudly, never silently) */

/* ---- constants of the path -------------------------------------------- */
#define LCTY_GC_BINS        101    /* src/bg/depth.rs:42 */
#define LCTY_DEPTH_CACHE    256    /* src/model/distr_cache.rs:14 */
#define LCTY_MAX_ALT_CN     15     /* src/math/distr/bayes.rs:4,16 (alternatives.len() < 16) */
#define LCTY_MAX_USED_ALNS  10     /* src/model/locs.rs:743 */
#define LCTY_MAX_UNUSED_ALNS 2     /* src/model/locs.rs:740 */
#define LCTY_NONE_U32       0xFFFFFFFFu

/* sequencing technology (src/bg/mod.rs:182-268) */
#define LCTY_TECH_ILLUMINA 0
#define LCTY_TECH_HIFI     1
#define LCTY_TECH_PACBIO   2
#define LCTY_TECH_NANOPORE 3

/* edit-distance threshold kind (src/bg/err_prof.rs:365-399) */
#define LCTY_EDIT_FRACTION 0
#define LCTY_EDIT_PVALUE   1

/* read status after AllAlignments::load (src/model/locs.rs:1116-1150, 1255-1286) */
#define LCTY_READ_GOOD          0  /* goes to AllAlignments::reads */
#define LCTY_READ_POORLY_MAPPED 1
#define LCTY_READ_OUT_OF_BOUNDS 2
#define LCTY_READ_FEW_KMERS     3  /* goes to AllAlignments::unused_reads */

/* alignment record flags: BAM flag bits, plus MATE2 assigned by the loader
 * (src/model/locs.rs:1119-1131: first primary-led group = ReadEnd::First). */
#define LCTY_FLAG_UNMAPPED  0x0004u
#define LCTY_FLAG_REVERSE   0x0010u
#define LCTY_FLAG_MATE2     0x0080u
#define LCTY_FLAG_SECONDARY 0x0100u
#define LCTY_FLAG_SUPPL     0x0800u

/* raw BAM CIGAR op codes accepted on the path (src/seq/cigar.rs:116-129) */
#define LCTY_CIGAR_M 0u  /* rejected: the path requires --eqx (src/seq/aln.rs:311) */
#define LCTY_CIGAR_I 1u
#define LCTY_CIGAR_D 2u
#define LCTY_CIGAR_S 4u
#define LCTY_CIGAR_H 5u
#define LCTY_CIGAR_EQ 7u
#define LCTY_CIGAR_X 8u

/* ---- plain-data structs crossing the boundary -------------------------- */

/* model::Params (src/model/mod.rs:64-135) — the fields the path consumes. */
typedef struct lcty_params {
    uint32_t boundary_size;      /* 200 */
    int32_t  tweak;              /* -1 = auto (set_tweak_size, model/mod.rs:179-197) */
    double   lik_skew;           /* 0.85 */
    double   prob_diff;          /* NaN = auto: |unmapped_penalty| + ln 10 (command/genotype.rs:1294-1296) */
    double   unmapped_penalty;   /* NaN = auto by technology (model/mod.rs:55-60) */
    double   poor_compl;         /* 0.5 */
    double   poor_compl_edit;    /* 0.7 */
    double   compl_weight_bp;    /* 0.5; <= 0 disables the calculator (None) */
    double   compl_weight_pow;   /* 4 */
    double   kmers_weight_bp;    /* 0.2; <= 0 disables */
    double   kmers_weight_pow;   /* 4 */
    double   min_weight;         /* 0.001 */
    double   filt_diff;          /* ln 1e100 */
    double   prob_thresh;        /* ln 1e-4 */
    double   alt_cn[LCTY_MAX_ALT_CN]; /* 0.3,2,3,4,5 */
    uint32_t n_alt_cn;           /* 5 */
    uint16_t kmer_soft_thresh;   /* 5 */
    uint16_t kmer_hard_thresh;   /* 1 */
    uint8_t  complexity_k;       /* 5 */
    uint8_t  dont_skip;          /* 0 */
    uint8_t  strict_subset;      /* locs.rs:490: BAM header has fewer contigs than the contig set */
    uint8_t  _pad0;
    uint32_t threads;            /* reference `-@` (only enters truncate_ixs / discard thresholds) */
} lcty_params;

/* bg::BgDistr as loaded from distr.gz (src/bg/mod.rs:147-177) */
typedef struct lcty_bg {
    double   op_lnprobs[5];      /* matches, mismatches, insertions, deletions, clipping (bg/err_prof.rs:321-329) */
    double   edit_alpha;         /* BetaBinomial(alpha, beta) of the error profile */
    double   edit_beta;
    double   ins_n;              /* insert-size NBinom(n,p); ignored when !is_paired (bg/insertsz.rs:195-208) */
    double   ins_p;
    double   depth_n[LCTY_GC_BINS]; /* bg_depth NBinom per GC bin (bg/depth.rs:400-412) */
    double   depth_p[LCTY_GC_BINS];
    double   edit_p1;            /* EditThresh params: Fraction(.03,.06) / PValue(.99,.999) (bg/err_prof.rs:394-399) */
    double   edit_p2;
    uint32_t window;             /* bg_depth.window */
    uint32_t neighb;             /* bg_depth.neighb */
    int32_t  is_paired;          /* insert_distr defined */
    int32_t  technology;         /* LCTY_TECH_* */
    int32_t  edit_kind;          /* LCTY_EDIT_* */
    uint32_t _pad0;
} lcty_bg;

/* One BAM record of OUT/loci/<locus>/aln.bam reduced to what the path reads
 * (src/seq/aln.rs:147-157, src/seq/cigar.rs:203-208). 16 bytes. */
typedef struct lcty_aln_rec {
    uint32_t pos;        /* 0-based leftmost reference position (record.pos()) */
    uint16_t contig;     /* ContigId of the allele */
    uint16_t flags;      /* LCTY_FLAG_* */
    uint32_t n_cigar;    /* number of raw CIGAR words */
    uint32_t cigar_rel;  /* offset of the first CIGAR word relative to the pair's cigar_off */
} lcty_aln_rec;

/* Host view of a chunk of read pairs (or single reads) with all their records, in
 * the input-order contract of locs.rs:1116-1150: per pair
 *   mate-1 primary, mate-1 secondaries..., mate-2 primary, mate-2 secondaries...
 * Sequences are the primary record's SEQ (BAM orientation), 2-bit packed
 * (A=0,C=1,G=2,T=3; src/seq/kmers.rs:179-183) with a 1-bit/base "not ACGT" side
 * channel because kmers() emits UNDEF for such windows (kmers.rs:184-190). */
typedef struct lcty_reads_host {
    uint64_t n_pairs;
    const uint32_t*     mate_len;   /* [2*n_pairs]; 0 = mate absent (single-end data: every odd entry 0) */
    const uint64_t*     mate_off;   /* [2*n_pairs+1] base offsets, each a multiple of 32 */
    const uint32_t*     bases2;     /* 16 bases per word, base i of a mate at bit 2*(i%16) of word (off+i)/16 */
    const uint32_t*     nmask;      /* 32 bases per word, bit set = base is not A/C/G/T */
    const uint64_t*     aln_off;    /* [n_pairs+1] record offsets */
    const lcty_aln_rec* recs;
    const uint64_t*     cigar_off;  /* [n_pairs+1] CIGAR word offsets */
    const uint32_t*     cigar;      /* raw BAM CIGAR words: len<<4 | op */
} lcty_reads_host;

/* One alignment with its operations already counted — the 16-byte alignment-table entry of SURVEY.md section 8(d). It holds what
 * Alignment::count_region_operations_fast (src/seq/aln.rs:301-317) with limited_clipping (288-296) leaves of a record: the caller
 * (who walks raw_cigar() anyway and knows the contig length) counts, the device computes edit_distance (bg/err_prof.rs:73-79),
 * ErrorProfile::ln_prob (212-221) and everything behind. Counts are 16-bit: reads of up to 65 535 bases; longer reads, hard-clipped
 * primaries and the other checks of the record path (seq/aln.rs:311, model/locs.rs:526) stay with the caller, as does alignment
 * recovery (lcty_recover_alignments needs the CIGARs and refuses a counted batch). */
#define LCTY_CF_REVERSE     (1u << 28)
#define LCTY_CF_NOT_PRIMARY (1u << 29)   /* secondary or supplementary record */
#define LCTY_CF_UNMAPPED    (1u << 30)
typedef struct lcty_aln_counted {
    uint32_t pos_flags;   /* 0-based leftmost position (28 bits) | LCTY_CF_* */
    uint16_t contig;      /* ContigId of the allele */
    uint16_t matches, mismatches, insertions, deletions, clipping;   /* OperCounts (bg/err_prof.rs:25-45), clipping already limited */
} lcty_aln_counted;

/* PairAlignment (src/model/locs.rs:668-676); LCTY_NONE_U32 encodes Option::None. */
typedef struct lcty_pair_aln {
    double   ln_prob;    /* already multiplied by the read weight (locs.rs:861-863) */
    uint32_t ix1;        /* record index inside the pair, or NONE */
    uint32_t mid1;       /* Interval::middle of the mate-1 alignment, or NONE */
    uint32_t ix2;
    uint32_t mid2;
    uint16_t contig;
    uint16_t _pad[3];
} lcty_pair_aln;

/* Solver of one stage (src/solvers/stoch.rs): Greedy (36-145) or SimAnneal (151-266) with their parameters
 * (`-S greedy:x0=..,s=..,p=..` / `-S anneal:n=..,p=..,P=..`, SetParams 128-139, 249-260). */
#define LCTY_SOLVER_GREEDY 0
#define LCTY_SOLVER_ANNEAL 1
/* The exact solver in the place of the reference's ILP back ends (HiGHS: src/solvers/highs.rs:38-134, Gurobi: gurobi.rs:15-83;
 * registered at solve.rs:152-171, `-S highs` / `-S gurobi`): the same model — one binary per (non-trivial read, location), one-hot
 * depth variables per window, coupling rows, objective = ReadAssignment::likelihood — built on the device by the stage's
 * initialisation and solved by branch and bound ON THE HOST: the models of a stage's chains are copied to the host, solved by a pool
 * of host threads, one model per thread at a time (as the reference runs one model per worker, solve.rs:1052-1062;
 * lcty_ctx_set_knob "exact_threads", default: the machine's hardware threads, at most 64), and the assignments are written back
 * into the chains' device records. `node_limit` nodes without an answer -> LCTY_ERR_SOLVER, as a non-optimal solver status is
 * upstream (highs.rs:113-116). Every attempt starts from the best location of every read; with tweak = 0 the attempts of a genotype
 * share one model, which is solved once.
 * `init_prob` is this kind's relative gap g: the search stops when nothing left can beat the incumbent by more than g x |incumbent|.
 * Default 1e-4 = HiGHS' default mip_rel_gap, at which the reference's runs report "optimal" (highs.rs:103-110 changes no option);
 * 0 asks for a proof of optimality (reaches ~1 000 read pairs; larger loci end in LCTY_ERR_SOLVER). The bound is the relaxation of
 * that programme without cuts (~0.45 above the optimum HiGHS proves at its root node): at the default gap every genotype of a locus of
 * 10 000 read pairs x 8 alleles is answered within 3e-5 of HiGHS' optimum (tests/test_exact_highs.py), models of a few thousand read
 * pairs and fewer may end in LCTY_ERR_SOLVER. */
#define LCTY_SOLVER_EXACT 2
typedef struct lcty_solver {
    int32_t  kind;          /* LCTY_SOLVER_* */
    int32_t  best_start;    /* greedy: 1 = start from the best location of every read (default), 0 = random */
    uint32_t sample_size;   /* greedy: 10 */
    uint32_t plato_size;    /* greedy: 100; anneal: 10000 */
    uint32_t anneal_steps;  /* anneal: 20000 */
    uint32_t node_limit;    /* exact: branch-and-bound nodes per attempt before LCTY_ERR_SOLVER (default 20 000 000) */
    double   init_prob;     /* anneal: 0.5; exact: relative gap at which the search stops and the answer counts as optimal = HiGHS'
                             * mip_rel_gap, default 1e-4 as the reference's runs (highs.rs:103-110 leaves the option alone); 0 = a proof */
} lcty_solver;

typedef struct lcty_ctx   lcty_ctx;
typedef struct lcty_locus lcty_locus;
typedef struct lcty_reads lcty_reads;

/* ---- library / context -------------------------------------------------- */
const char* lcty_last_error(void);
const char* lcty_version(void);
/* number of HIP devices visible; 0 when there is none (never an error) */
int32_t lcty_device_count(void);
int32_t lcty_ctx_create(int32_t device_id, lcty_ctx** out);
void    lcty_ctx_destroy(lcty_ctx* ctx);
int32_t lcty_ctx_synchronize(lcty_ctx* ctx);
/* Limits of the retry / batching machinery of this context and choices between equivalent kernel forms, for tests that must reach
 * those paths with small inputs (no environment variable changes what the library computes):
 *   "transfer_levels", "transfer_scratch_mb", "transfer_waves", "transfer_cap_new", "transfer_arena"   lcty_recover_alignments: scratch
 *       levels, arenas;
 *   "depth_table_start"   first width of the extended depth table;   "solve_budget_mb"   device memory for the per-chain state of a
 *       solver stage;   "solve_extra_start"   first size of a chain's run of locations beyond the second;
 *   "solve_chains_per_wave"   1, 2, 4, 5, 6 chains of the greedy loop per wavefront;   "solve_lds_weights"   0: the greedy loop gathers
 *       the window weights, 1 (default where the locus has the weight tables): table indices + tables in LDS;
 *   "anneal_lds_weights"   0 gathered, 1 in LDS as they are, 2 (default where possible) table indices + tables in LDS;
 *   "contig_info_slide"   0 / 1: lcty_locus_create counts its neighbourhoods position by position / with sliding windows (default:
 *       sliding from 1 024 bases on);   "solve_stats"   1: per-stage iteration counts on stderr;   "queue_trace"   1: wall-clock marks of the phases of every locus of
 *       lcty_solve / lcty_solve_queue on stderr;   "gather_chunk_mb"   staging size of lcty_solve_stage_read_sharded;
 *   "prefilter_gram"   0: always the f64 tile kernel, 1: the integer Gram contraction on the matrix cores whenever it applies
 *       (default: from 512 alleles on);   "arena_cap_pct"   p: batches created afterwards (lcty_reads_create) get p % of the bound on their PairAlignment arena (two per
 *       record; one per (pair, allele) is the rule — an arena that is too small fails loudly);   "score_lean"   0: counted batches go through the general scoring kernel only (default 1: the lean kernel first, the general
 *       one on the pairs it leaves);   "score_lean_keep"   0: the lean kernel scores a saved record again in its last pass instead of
 *       keeping the first pass' products in LDS (what loci of more than 310 alleles get anyway);   "score_timing"   1: the lean kernel's
 *       timed build, shader-clock ticks per phase on stderr;   "comm_fail_at"   k: the k-th status agreement of a multi-GPU call fails on this rank (tests of
 *       the error path of the exchanges);   "prefilter_gram_cols"   room for that many level columns per read (default 6; too few: the
 *       f64 kernel takes the batch);   "queue_early_head"   0: lcty_solve_queue makes the head of a locus after the chains of the locus before (default 1:
 *       beside them);   "prefilter_gram_levels"   levels of a row the contraction takes (<= 16; rows with more go
 *       through the f64 kernel).
 * value < 0 restores the default; an unknown name is LCTY_ERR_INVALID_INPUT. None of them changes a result beyond the last bits of
 * an f64 sum (the order in which a chain's likelihood or a genotype's score is added up). */
int32_t lcty_ctx_set_knob(lcty_ctx* ctx, const char* name, int64_t value);
/* Files a developer asks the library to write (no environment variable is read anywhere in the library): name "exact_dump" = the
 * model of the first chain of an exact-solver stage as text, written when that stage runs; path NULL or "" switches it off.
 * No counterpart upstream (HiGHS' own `write_model` is not called by highs.rs). */
int32_t lcty_ctx_set_path(lcty_ctx* ctx, const char* name, const char* path);
/* The solver stages keep their per-chain device state (32 B per chain and good read pair: ~150 GB for the 5 000 greedy chains of
 * the default scheme at 1 M read pairs; batches of chains when the device has less) with the context between stages and loci;
 * this releases it (the next stage allocates again). So does alignment recovery with its lane scratch and the arenas of the
 * transferred alignments (up to half of what is free on the device when lcty_recover_alignments first runs; long reads need tens of MB
 * per wavefront): kept between calls, handed back when a solver stage sizes its workspace, and by this call. */
int32_t lcty_ctx_trim(lcty_ctx* ctx);
/* Page-locked host memory for the chunks handed to lcty_reads_append / lcty_reads_append_counted / lcty_recruit_*: the copies of
 * a chunk that lies in such memory go over PCIe at link rate (a pageable chunk is staged through the driver's bounce buffers at
 * about half of it). The reference reads its records into ordinary Vecs (locs.rs:1116-1150); a binding would fill these
 * buffers instead. Release with lcty_host_free. */
int32_t lcty_host_alloc(lcty_ctx* ctx, uint64_t bytes, void** out);
void    lcty_host_free(void* p);

/* defaults of model::Params::default (model/mod.rs:108-135) */
void    lcty_params_default(lcty_params* out);
/* Resolves the "auto" fields exactly as command/genotype.rs:1282-1296 does:
 * tweak (model/mod.rs:179-197), unmapped_penalty (55-60), prob_diff. */
int32_t lcty_params_resolve(lcty_params* params, const lcty_bg* bg);

/* ---- locus: ContigSet + KmerCounts + ContigInfos + UniqueKmers + LUTs ----
 * Replaces ContigSet::load_with_kmer_counts (src/seq/contigs.rs:295),
 * ContigInfos::new (src/model/windows.rs:584-615), UniqueKmers::new
 * (src/model/locs.rs:930-963), InsertDistr::load (src/bg/insertsz.rs:195-208),
 * EditDistCache::new (src/bg/err_prof.rs:422-428), DistrCache::new
 * (src/model/distr_cache.rs:61-75).
 *   seqs       ASCII allele sequences, concatenated; seq_off[n_alleles+1]
 *   offtarget  off-target k-mer counts (first KmerCounts block), cnt_off[n_alleles+1],
 *              cnt_off[a+1]-cnt_off[a] == len(a)+1-k
 *   k          2 <= k <= 63 as in the reference (u128 k-mers, src/seq/kmers.rs:8-26; locs.rs:919). k <= 31 (the default of
 *              `locityper add` is 25): 64-bit keys, the set built on the device, the reads' k-mers taken from registers;
 *              32 <= k <= 63: 128-bit keys in a table of {lo, hi} pairs built on the host, the reads' k-mers from memory          */
int32_t lcty_locus_create(lcty_ctx* ctx, uint32_t n_alleles,
                          const uint8_t* seqs, const uint64_t* seq_off,
                          const uint16_t* offtarget, const uint64_t* cnt_off, uint32_t k,
                          const lcty_bg* bg, const lcty_params* params, lcty_locus** out);
void    lcty_locus_destroy(lcty_locus* locus);
/* KmerCounts::load (src/seq/counts.rs:127-150) on the decompressed bytes of `kmers.bin.br` / `.lz4` (command/paths.rs:4-5): parses
 * the FIRST block, which holds the off-target counts (command/add.rs:647-650; contigs.rs:301-303), into the arrays
 * lcty_locus_create takes. Values are clamped to min(65535, 2^(8 * counter bytes) - 1). Two calls: with cnt_off = counts = NULL it
 * only reports k, the number of contigs and (in *consumed) the bytes of the block; the total number of values is then
 * cnt_off[n_contigs] of the second call, for which cap_counts may be the remaining length of the buffer (a value takes at least
 * one byte). Truncated data / overlong varints / a counter length > 8: LCTY_ERR_INVALID_DATA. Host code. */
int32_t lcty_kmer_counts_parse(const uint8_t* buf, uint64_t len, uint32_t* k, uint32_t* n_contigs, uint64_t* cnt_off /* [n_contigs+1] */,
                               uint64_t cap_contigs, uint16_t* counts, uint64_t cap_counts, uint64_t* consumed);
/* number of locus-unique canonical k-mers (the count logged at locs.rs:953) */
int32_t lcty_locus_n_unique_kmers(const lcty_locus* locus, uint64_t* out);
/* ContigInfo::new products per allele (windows.rs:386-407): position count = len-neighb+1.
 * Any output pointer may be NULL. uniq counts / complexity counts are the integer
 * numerators, the f64 values of the reference are count*mult with
 *   uniq_kmer_frac = uniq * (1/(neighb+1-k)),  complexity = distinct * (1/min(neighb+1-ck, 4^ck)). */
int32_t lcty_locus_contig_info(const lcty_locus* locus, uint32_t allele,
                               uint8_t* gc, uint32_t* uniq_cnt, uint16_t* compl_cnt,
                               uint32_t* n_windows, uint32_t* reg_start);
/* EditDistCache::get (bg/err_prof.rs:434-448) */
int32_t lcty_locus_edit_thresholds(const lcty_locus* locus, uint32_t read_len, uint32_t* good, uint32_t* passable);
/* InsertDistr::ln_prob / insert_penalty (bg/insertsz.rs:153-175) */
int32_t lcty_locus_insert_lnprob(const lcty_locus* locus, uint32_t n, const uint32_t* sizes, double* out, double* insert_penalty);
/* DistrCache: ln P(depth) for gc bin, depth in 0..LCTY_DEPTH_CACHE (distr_cache.rs:61-75; bayes.rs:27-35) */
int32_t lcty_locus_depth_lut(const lcty_locus* locus, double* out /* [101*256] */);
/* ContigInfo::neighb_info weight (windows.rs:439-445) of every moving-window position, alleles concatenated
 * (sum over alleles of len - neighb + 1 values) */
int32_t lcty_locus_window_weights(const lcty_locus* locus, double* out);
/* The depth table of the solver stages: DistrCache (src/model/distr_cache.rs:61-92) past the 256 entries of the reference's
 * LinearCache, the same BayesCalc::ln_pmf (src/math/distr/bayes.rs:27-35) evaluated on the device. *width_io is rounded up to a
 * power of two >= 256; out (may be NULL to ask for the width) receives [101][width]. */
int32_t lcty_locus_depth_table(lcty_locus* locus, uint32_t* width_io, double* out);

/* Explicit region weights (`locityper genotype --reg-weights`): replaces load_explicit_weights (src/model/windows.rs:257-317)
 * minus the text parsing — the lines of the BED file in file order as (allele index, start, end, value); an index >= n_alleles
 * stands for a contig name the locus does not have (the line is skipped, 269-272). Errors as upstream: value outside [0, 1],
 * an allele not covered from its first base on without gaps, missing, or covered to a different length -> LCTY_ERR_INVALID_DATA
 * (Error::ParsingError); an interval beyond the end of its allele -> LCTY_ERR_INVALID_INPUT (seq/interv.rs:112-116).
 * Effects, both on the device: every window weight gets the window's average as its last factor (ExplicitWeights::average,
 * 236-238; ContigInfo::new 409-413; neighb_info 441-443), and lcty_score_reads multiplies a read pair's weight by
 * ContigInfos::explicit_read_weight over its PairAlignments (683-693, read_end_weight 493-503; locs.rs:860, 903).
 * A NaN value is rejected as out of range (upstream's `val < 0.0 || val > 1.0` lets it through and every weight turns NaN).
 * Call it after lcty_locus_create and before the reads of the locus are scored; a second call replaces the first.
 * (ContigInfos::weighted_aln_prob / average_read_weight, windows.rs:506-643, have no caller upstream and are not built.) */
int32_t lcty_locus_set_explicit_weights(lcty_locus* locus, uint32_t n_lines, const uint32_t* allele, const uint32_t* start,
                                        const uint32_t* end, const double* value);

/* ---- reads: device-resident batch -----------------------------------------
 * Capacity is fixed at creation so a batch larger than host memory can be
 * appended chunk by chunk (each append is one set of H2D copies).             */
int32_t lcty_reads_create(lcty_locus* locus, uint64_t cap_pairs, uint64_t cap_bases,
                          uint64_t cap_recs, uint64_t cap_cigar, lcty_reads** out);
/* A batch larger than HBM (1 M ONT reads x 256 alleles carry 600 GB of CIGAR words): the records, CIGAR words and bases of ONE
 * chunk are on the device at a time, the products of every scored chunk stay (status, weights, k-mer counts, matrix rows,
 * PairAlignments: about 9 KB per read at 256 alleles). Use:  create_streaming; { lcty_reads_append (one or more chunks that fit
 * the chunk capacities); lcty_score_reads } ...; then prefilter / solver stages / read-backs as for any batch — they see all
 * pairs appended so far, in order, and the results are those of one resident batch (AllAlignments::load is a loop over reads,
 * locs.rs:1119-1185). An append after a score drops the records of the scored chunk. cap_pair_alns = room for that many
 * PairAlignments in total (0: three per (pair, allele)); overflow fails loudly. With alignment recovery the loop body is
 * { append; lcty_score_reads; lcty_recover_alignments; lcty_score_reads }: recovery works on the chunk whose records are resident. */
int32_t lcty_reads_create_streaming(lcty_locus* locus, uint64_t cap_pairs, uint64_t chunk_pairs, uint64_t chunk_bases,
                                    uint64_t chunk_recs, uint64_t chunk_cigar, uint64_t cap_pair_alns, lcty_reads** out);
int32_t lcty_reads_append(lcty_reads* reads, const lcty_reads_host* chunk);
/* The same chunk with its records counted by the caller: alns[aln_off[n_pairs]] in the record order of the chunk; chunk->recs,
 * cigar_off and cigar are not read. A batch holds records or counted alignments, never both. lcty_score_reads, the prefilter and
 * the solver stages see no difference; results equal the record path's bit for bit (tests/test_gpu_counted.py). */
int32_t lcty_reads_append_counted(lcty_reads* reads, const lcty_reads_host* chunk, const lcty_aln_counted* alns);
/* An empty batch again, bound to `locus` (same context, not more alleles than the locus it was created for): every buffer stays.
 * For queues of loci that rotate over a few batch objects — allocating or releasing tens of GB per locus would wait for every
 * stream of the device. Nothing of the batch may be in use (see lcty_solve_queue_fed's release). */
int32_t lcty_reads_reset(lcty_reads* reads, lcty_locus* locus);
void    lcty_reads_destroy(lcty_reads* reads);
int32_t lcty_reads_n_pairs(const lcty_reads* reads, uint64_t* out);

/* AllAlignments::load without alignment recovery (src/model/locs.rs:1085-1185,
 * 1237-1288 with opt_hap_alns == None): K2 unique k-mers + read weight (968-1002),
 * K4 op counts + ErrorProfile::ln_prob (aln.rs:301-317, err_prof.rs:212-221),
 * K5 thresholds / 128-bp dedupe (502-567, 298-344), K7 pairing (746-868 / 873-911)
 * and K8 the dense row of best_aln_matrix (1203-1212) — one fused launch.
 * Asynchronous on the context's stream.                                         */
int32_t lcty_score_reads(lcty_reads* reads);

/* per-pair products of load(): any pointer may be NULL */
int32_t lcty_reads_get_status(lcty_reads* reads, uint8_t* status, double* weight,
                              double* unmapped_prob, uint16_t* uniq_kmers /* [2*n_pairs] */);
/* number of LCTY_READ_GOOD pairs (AllAlignments::reads().len()) */
int32_t lcty_reads_n_good(lcty_reads* reads, uint64_t* out);
/* AllAlignments::best_aln_matrix (locs.rs:1203-1212): out[a*n_good + j], j over GOOD
 * pairs in input order (the order produced with threads == 1, locs.rs:1149). */
int32_t lcty_best_aln_matrix(lcty_reads* reads, double* out);
/* GrouppedAlignments::aln_pairs of every GOOD and FEW_KMERS pair, CSR over all input
 * pairs: off[n_pairs+1]; entries contig-ascending, ln_prob-descending inside a contig
 * (locs.rs:819-851). Call with out == NULL to obtain only the offsets / total. */
int32_t lcty_reads_get_pair_alns(lcty_reads* reads, uint64_t* off, lcty_pair_aln* out, uint64_t cap);
/* The record table as the batch holds it now: after lcty_recover_alignments the caller's records with the transferred alignments
 * behind the original records of their read end (secondary records, CIGAR offsets relative to the pair's block as everywhere) — the
 * `table` lcty_write_bam needs then. aln_off / cigar_off [n_pairs + 1]; recs = cigar = NULL: the sizes only (aln_off[n_pairs],
 * cigar_off[n_pairs]). Not for counted or streaming batches. */
int32_t lcty_reads_get_records(lcty_reads* reads, uint64_t* aln_off, lcty_aln_rec* recs, uint64_t cap_recs, uint64_t* cigar_off, uint32_t* cigar,
                               uint64_t cap_cigar);

/* run_filter (src/solvers/solve.rs:87-122): scores[g] = prior[g] + sum_r max_{a in g} M[a][r].
 * genotypes == NULL: all multisets of size `ploidy` in the order of
 * gen_combinations_with_repl (src/ext/vec.rs:298-339), n_genotypes is then checked
 * against count_combinations_with_repl. priors == NULL: all 0.0.                     */
int32_t lcty_prefilter(lcty_reads* reads, const uint16_t* genotypes, uint64_t n_genotypes,
                       uint32_t ploidy, const double* priors, double* scores);
/* device-only variant used by the timed path: leaves the scores in HBM */
int32_t lcty_prefilter_async(lcty_reads* reads, uint32_t ploidy);
int32_t lcty_prefilter_scores(lcty_reads* reads, double* scores, uint64_t n);

/* truncate_ixs (src/solvers/solve.rs:52-84). ixs: in = candidate indices, out = kept,
 * sorted by (score desc, index asc); returns the kept count in *n_keep.              */
int32_t lcty_truncate(const double* scores, uint64_t* ixs, uint64_t n, double filt_diff,
                      uint64_t min_size, uint64_t threads, uint64_t* n_keep);

/* The same on the scores the last prefilter call left on the device (lcty_prefilter_async / lcty_prefilter /
 * lcty_prefilter_allreduce), all genotypes as candidates: a stable device radix sort of (score descending, index ascending) and the
 * prefix truncate_ixs keeps; only the kept indices come to the host (at 4 096 alleles the scores are 67 MB). ixs = NULL: *n_keep only. */
int32_t lcty_prefilter_truncate(lcty_reads* reads, double filt_diff, uint64_t min_size, uint64_t threads, uint64_t* ixs, uint64_t cap,
                                uint64_t* n_keep);
/* scores[g] = priors[g] + scores[g] on the device (run_filter, solve.rs:114: `--priors`), between lcty_prefilter_async (or the
 * all-reduce) and lcty_prefilter_truncate; n = the number of genotypes of the last prefilter call. */
int32_t lcty_prefilter_add_priors(lcty_reads* reads, const double* priors, uint64_t n);

/* generate_genotypes without priors (src/command/genotype.rs:1120-1126) */
uint64_t lcty_count_genotypes(uint32_t n_alleles, uint32_t ploidy);
int32_t  lcty_generate_genotypes(uint32_t n_alleles, uint32_t ploidy, uint16_t* out, uint64_t cap);

/* Per-read posteriors: assignment counts of a genotype's attempts (lcty_assignment_counts) -> probability and mapping quality of
 * every (read, location), as the output BAMs carry them in the `pr` tag / MAPQ — count_to_prob (src/model/bam.rs:56-67):
 * 0 -> (0, 0); all attempts -> (1, 60); otherwise count / attempts in f32 and min(60, round(-10 log10(1 - p))). */
int32_t lcty_counts_to_posteriors(const uint16_t* counts, uint64_t n, uint16_t attempts, float* prob, uint8_t* mapq);

/* ---- one locus over several GPUs: the exchange step -----------------------------------------------------------
 * run_filter's score of a genotype is a sum over reads (solve.rs:105-119): with the reads of a locus sharded over ranks (one process
 * per GPU), every rank runs lcty_score_reads + lcty_prefilter_async on its shard and lcty_prefilter_allreduce sums the G-long f64
 * vectors in place on the devices (RCCL all-reduce over xGMI); lcty_prefilter_scores then returns the scores of the whole batch on
 * every rank and truncate_ixs proceeds identically everywhere. Priors are added afterwards (they are not per-read).
 * lcty_comm_unique_id: ncclGetUniqueId on rank 0; the launcher hands the 128 bytes to the other ranks. n_ranks == 1 is allowed. */
#define LCTY_COMM_ID_BYTES 128
typedef struct lcty_comm lcty_comm;
int32_t lcty_comm_unique_id(uint8_t* id);
int32_t lcty_comm_create(lcty_ctx* ctx, int32_t n_ranks, int32_t rank, const uint8_t* id, lcty_comm** out);
void    lcty_comm_destroy(lcty_comm* comm);
/* what RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank): the launcher's proof of how many ranks an
 * exchange really spans (bench.py prints it) */
int32_t lcty_comm_ranks(const lcty_comm* comm, int32_t* n_ranks, int32_t* rank);
/* Every exchange below: whatever a rank does on its own between two collectives (checks, allocations, launches) runs behind a
 * status agreement — a three-word MAX all-reduce every rank joins unconditionally — so that a failure on one rank makes ALL ranks
 * return an error (the failing rank its own status, the others "another rank failed") instead of leaving them inside RCCL; the
 * same agreement carries the sizes the next collective depends on and refuses ranks that disagree about them.
 * (Test hook: lcty_ctx_set_knob "comm_fail_at" = k fails the k-th agreement of a call on this rank.) */
int32_t lcty_prefilter_allreduce(lcty_reads* reads, lcty_comm* comm);
/* One solver stage with its (genotype, attempt) chains dealt to the ranks of `comm` (SURVEY.md 8e level 3; the reference deals
 * the genotypes of a stage to its worker threads, solve.rs:1052-1062): every rank passes the SAME arguments and holds the same
 * scored reads; rank r runs the r-th contiguous block of the genotype list, the per-chain likelihoods are all-gathered on the
 * devices (RCCL) and every rank returns mean / variance / likelihoods of ALL n_gt genotypes — bit for bit what lcty_solve_stage
 * returns on one GPU, for any number of ranks. */
int32_t lcty_solve_stage_sharded(lcty_reads* reads, lcty_comm* comm, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy,
                                 const double* priors, const lcty_solver* solver, uint32_t attempts, const uint64_t* chain_seeds,
                                 double* lik_mean, double* lik_var, double* liks_out /* [n_gt][attempts] or NULL */);
/* One solver stage of a locus whose READS are sharded over the ranks (SURVEY.md 8e level 2 carried through the solver; BASELINE
 * configs[4]): rank r scored the r-th contiguous block of the locus' read list into `shard`. GenotypeAlignments::new walks every
 * read of the locus (assgn.rs:41-84), so the ranks exchange what a stage needs of it: the location-table rows of the alleles
 * of the stage's genotypes — 32 B per (allele, good read pair) plus the runs of further pair-alignments — packed, all-gathered
 * in chunks of rows (RCCL) and laid side by side in rank order. The chains are then dealt to the ranks as in
 * lcty_solve_stage_sharded and their likelihoods all-gathered. Every rank passes the same stage arguments and gets the results of
 * all n_gt genotypes — bit for bit what lcty_solve_stage returns on the unsharded batch. Device memory: alleles-of-the-stage x
 * good read pairs of the locus x 32 B per rank (LCTY_ERR_RUNTIME when that does not fit).
 * lcty_count_unexplained of a sharded locus: the sum of the ranks' counts. */
int32_t lcty_solve_stage_read_sharded(lcty_reads* shard, lcty_comm* comm, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy,
                                      const double* priors, const lcty_solver* solver, uint32_t attempts, const uint64_t* chain_seeds,
                                      double* lik_mean, double* lik_var, double* liks_out /* [n_gt][attempts] or NULL */);
/* The same with every shard on ONE device and no exchange: `shards` in read order, all of one locus and one context. The packing
 * and the side-by-side layout are the code lcty_solve_stage_read_sharded runs between its collectives. Scratch lives on shards[0]. */
int32_t lcty_solve_stage_from_shards(lcty_reads* const* shards, uint32_t n_shards, const uint16_t* genotypes, uint64_t n_gt,
                                     uint32_t ploidy, const double* priors, const lcty_solver* solver, uint32_t attempts,
                                     const uint64_t* chain_seeds, double* lik_mean, double* lik_var, double* liks_out);

/* ---- alignment recovery (AllAlignments::load with opt_hap_alns = Some; src/seq/transfer.rs, src/seq/cigar.rs:1085-1384,
 * src/seq/wfa.rs) ------------------------------------------------------------------------------------------------------
 * lcty_locus_set_hap_alns: the pairwise haplotype alignments of `haplotypes.paf` as HapAlns::add takes them (transfer.rs:41-62):
 * entry t aligns contig id1[t] (query) to contig id2[t] (target) over their full lengths on the forward strand
 * (PafEntry::full_positive_alignment), CIGAR = raw BAM words cigar[cigar_off[t] .. cigar_off[t+1]) with =, X, I, D;
 * n_matches / aln_len as in the PAF columns (divergence filter: (aln_len - n_matches) / aln_len <= max_div, paf.rs:201-208).
 * The first entry of a pair of contigs wins; targets of a contig are tried in order of decreasing n_matches.
 * lcty_recover_alignments: transfer_alignments (transfer.rs:70-140) for every read pair of the batch that reaches
 * recover_and_group_alignments with weight >= min_weight (locs.rs:1255-1260). Call order:
 *     lcty_score_reads -> lcty_recover_alignments -> lcty_score_reads.
 * Transferred alignments become records of the batch (after the original records of their read end); nothing can be appended
 * to the batch afterwards. The aligner is an exact gap-affine dynamic programme (WFA2-lib computes the same optimum). Lanes
 * hold stretches between anchors of up to 255 bases; a read pair with a transfer that needs more is repeated with larger lane
 * scratch (2 047, then 16 383 bases; transferred CIGARs of up to 16 x the level-0 capacity) and LCTY_ERR_UNSUPPORTED beyond.
 * lcty_recover_stats: level_pairs[3] = read pairs the last lcty_recover_alignments took at each of the three levels. */
int32_t lcty_locus_set_hap_alns(lcty_locus* locus, uint32_t n_entries, const uint32_t* id1, const uint32_t* id2, const uint64_t* cigar_off,
                                const uint32_t* cigar, const uint32_t* n_matches, const uint32_t* aln_len, uint32_t transfer_fails,
                                double max_div);
int32_t lcty_recover_alignments(lcty_reads* reads, uint64_t* n_recovered);
int32_t lcty_recover_stats(lcty_reads* reads, uint64_t* level_pairs);
/* cells of the gap-affine aligner's matrices (the stand-in for WFA2, src/seq/wfa.rs:162-365) filled by the last
 * lcty_recover_alignments on this batch: cells / kernel time = the GCUPS figure of SURVEY.md section 8(d) */
int32_t lcty_recover_dp_cells(lcty_reads* reads, uint64_t* cells);

/* ---- minimizer read recruitment: the step of `locityper genotype` immediately before the path (SURVEY.md §8f rank 1;
 * src/seq/recruit.rs, src/seq/kmers.rs:71-340, src/math/frac.rs) ----------------------------------------------------------------
 * lcty_recruit_params_default: DEFAULT_MINIM_KW = (15, 10), match length 2000, k-mer threshold 50 (genotype.rs:136-139) and
 *   Technology::default_match_frac (bg/mod.rs:245-252).
 * lcty_targets_create / _add_locus / _finalize: recruit::Params::new (recruit.rs:65-105), TargetBuilder::add (688-738: canonical
 *   minimizers of every allele, rare = off-target count of the k-mer around the minimizer < thresh_kmer_count) and ::finalize.
 *   counts / cnt_off / base_k as in lcty_locus_create. Loci are numbered in the order they are added.
 * lcty_recruit: Targets::recruit_read_pair (883-929) for every pair of the chunk when `paired`, otherwise recruit_short_read
 *   (848-879) for single reads of up to 500 bases and recruit_long_read (938-996) beyond, as upstream dispatches (589-595).
 *   Mates of a pair: up to 256 bases; loci per read: up to 8 (16 for single reads beyond 256 bases); anything else is
 *   LCTY_ERR_UNSUPPORTED, never a different answer. Only the sequence fields of the chunk are read. Output: out_cnt[i] loci of
 *   pair i in out_loci[i * max_out ...], increasing (the reference writes the read to the files of exactly these loci). */
typedef struct lcty_recruit_params {
    double   match_frac;
    uint32_t match_length;
    uint16_t thresh_kmer_count;
    uint8_t  minimizer_k, minimizer_w;
} lcty_recruit_params;
typedef struct lcty_targets lcty_targets;
int32_t lcty_recruit_params_default(lcty_recruit_params* p, int32_t technology, int32_t is_paired);
int32_t lcty_targets_create(lcty_ctx* ctx, const lcty_recruit_params* params, lcty_targets** out);
void    lcty_targets_destroy(lcty_targets* targets);
int32_t lcty_targets_add_locus(lcty_targets* targets, uint32_t n_alleles, const uint8_t* seqs, const uint64_t* seq_off,
                               const uint16_t* counts, const uint64_t* cnt_off, uint32_t base_k, uint32_t* locus_ix);
int32_t lcty_targets_finalize(lcty_targets* targets, uint64_t* n_minimizers);
int32_t lcty_recruit(lcty_targets* targets, const lcty_reads_host* chunk, int32_t paired, uint32_t max_out, uint32_t* out_cnt,
                     uint32_t* out_loci);

/* ---- the readers and writers around recruitment (host; src/seq/fastx.rs, src/seq/recruit.rs:1000-1030) -------------------------
 * lcty_fastx_open: Reader::from_path (fastx.rs:296-312) on one FASTA / FASTQ file (plain or gzip), PairedEndInterleaved (444-466)
 *   with `interleaved`, PairedEndReaders (473-511) with a second file. lcty_fastx_next: up to max_records records (read pairs when
 *   paired) as the sequence fields of a chunk — what lcty_recruit reads; every byte but A, C, G, T is "not ACGT" (seq/kmers.rs:178-190)
 *   —; the arrays belong to the handle until the next call; *n = 0 at the end of the input. A record's name is its header up to the
 *   first blank (fastx.rs:417-419). Errors as upstream, LCTY_ERR_INVALID_DATA with the reference's messages: "Fastq record .. is
 *   incomplete" / "has incorrect format" / "has non-matching sequence and qualities", "Fasta record .. has an empty sequence.", "Odd
 *   number of records in an interleaved input file(s)", "non matching first and second mate(s)", "Different number of records in
 *   paired-end input files".
 * lcty_fastx_writers_open / lcty_fastx_write_recruited / _close: the per-locus read files of recruit_single_thread's loop
 *   (recruit.rs:1017-1021: `record.write_to(writers.get(locus_ix))` for every locus of the answer): record i of the LAST chunk goes
 *   to the writers out_loci[i * max_out .. + out_cnt[i]] as lcty_recruit filled them, as write_fastq / write_fasta leave it
 *   (fastx.rs:46-75) and both mates of a pair one after the other (141-150) — the `reads.fq` the mapper is given WITHOUT
 *   --interleaved. A path that ends in .gz is gzip-compressed. */
typedef struct lcty_fastx lcty_fastx;
typedef struct lcty_fastx_writers lcty_fastx_writers;
int32_t lcty_fastx_open(const char* path1, const char* path2, int32_t interleaved, lcty_fastx** out);
void    lcty_fastx_close(lcty_fastx* f);
int32_t lcty_fastx_is_paired(const lcty_fastx* f, int32_t* paired);
int32_t lcty_fastx_next(lcty_fastx* f, uint64_t max_records, lcty_reads_host* view, uint64_t* n);
int32_t lcty_fastx_writers_open(const char* const* paths, uint32_t n, lcty_fastx_writers** out);
int32_t lcty_fastx_write_recruited(lcty_fastx* f, lcty_fastx_writers* writers, uint32_t max_out, const uint32_t* out_cnt, const uint32_t* out_loci,
                                   uint64_t* n_written);
int32_t lcty_fastx_writers_close(lcty_fastx_writers* writers);

/* ---- candidate generation inside a locus (first slice) ----------------------------------------------------
 * The reference runs an external mapper per locus — strobealign -k 15 -N/-M min(25000, 4 x alleles) -S 0.5 --eqx for short
 * reads (src/command/genotype.rs:962-1005), piped through samtools view -e "[AS] >= 50 || flag & 2304 == 0" (1055-1094), on the
 * basis haplotypes when --basis is given (1007-1052) — and reads the resulting aln.bam. No mapper source is in the reference
 * tree; the algorithm of this slice is this build's own (locityper_amd/csrc/lcty_map.hip states it, tests/pyref_map.py restates
 * it): k-mer seeds every `stride` bases -> votes for (basis allele, strand, diagonal) -> per (allele, strand) the diagonal with
 * the most votes -> extension without gaps (+match / -mismatch per base, end_bonus per read end reached; the best-scoring stretch,
 * the rest soft-clipped), and for a clipped candidate a gap-affine alignment in a band around its diagonal that replaces it when it
 * scores higher -> the best candidate of a read end is its primary record, the others with score >= min_score secondary
 * records, a read end without candidates an unmapped record. Record order, flags, =/X/S CIGARs and SEQ orientation are those of
 * the BAM the reference reads, so the result is a chunk for lcty_reads_append; the alleles outside the basis are reached with
 * lcty_recover_alignments. Limits of this SHORT route: read ends of up to 256 bases, up to 32 basis alleles, seed length 8..31, at
 * most 64 seeds per read end and 1 024 votes (the first ones in seed order).
 * The LONG route (locityper_amd/csrc/lcty_map_long.hip states it, tests/pyref_map_long.py restates it) takes what the short one
 * refuses — the reference's long-read case, minimap2 -x map-ont / map-hifi --eqx (genotype.rs:990-1002): read ends of up to 2^20 - 1
 * bases on up to 256 basis alleles. Seeds as above (any number of them) -> every (seed, place) is an anchor (q, t) of its (basis allele,
 * strand), t on the allele in the read's orientation -> per (allele, strand) one chain: an anchor follows one of the `chain_back`
 * anchors of its group before it, at most `chain_gap` bases on in both sequences, at most `chain_skew` diagonals apart, gaining
 * min(dq, dt, k) - (0 if on the same diagonal else 2 + skew) -> the (allele, strand)s whose best chain has >= min_votes anchors and
 * >= half the score of the read end's best one are aligned along their chain: gap-affine (gaps open from any state) from node to
 * node, in a band of min(0, d) - band .. max(0, d) + band diagonals between two anchors d diagonals apart, and +- band beyond the first
 * and last anchor, where the alignment may stop (soft clip; end_bonus when the read end is reached) -> records as above, = / X / I / D / S
 * CIGARs. route: LCTY_MAP_ROUTE_AUTO takes the short route when the chunk and the index fit it.
 * lcty_locus_build_map_index: the k-mers of the basis alleles (hash table built on the device, once per locus).
 * lcty_map_reads: only the sequence fields of `chunk` are read. aln_off / cigar_off [n_pairs + 1] are always written; with
 *   recs == NULL the call only sizes. bases2_out / nmask_out: the chunk's bases in BAM orientation (same offsets). */
typedef struct lcty_map_params {
    uint32_t k, stride, min_votes;
    uint32_t max_occ;            /* seeds with more places in the index do not vote; 0: four per basis allele */
    int32_t  match, mismatch, end_bonus, min_score;
    uint32_t band;               /* diagonals on either side in the alignment with gaps of a clipped candidate (<= 16); 0: none */
    int32_t  gap_open, gap_extend;   /* a gap of n bases costs gap_open + (n - 1) * gap_extend */
    uint32_t route;              /* LCTY_MAP_ROUTE_* */
    uint32_t chain_gap;          /* long route: bases between two chained anchors, in either sequence (<= 8 192) */
    uint32_t chain_skew;         /* long route: diagonals between two chained anchors (<= 1 024) */
    uint32_t chain_back;         /* long route: anchors of its group an anchor looks back at (1..64) */
} lcty_map_params;
#define LCTY_MAP_ROUTE_AUTO  0u
#define LCTY_MAP_ROUTE_SHORT 1u
#define LCTY_MAP_ROUTE_LONG  2u
int32_t lcty_map_params_default(lcty_map_params* p);       /* strobealign's scores (short reads) */
int32_t lcty_map_params_default_long(lcty_map_params* p);  /* minimap2's (map-ont): seeds 16 bases apart, 2 / 4 / gap 4 + 2 n, every record kept */
int32_t lcty_locus_build_map_index(lcty_locus* locus, const uint16_t* basis, uint32_t n_basis, uint32_t k);
int32_t lcty_map_reads(lcty_locus* locus, const lcty_reads_host* chunk, const lcty_map_params* params, uint64_t* aln_off,
                       lcty_aln_rec* recs, uint64_t cap_recs, uint64_t* cigar_off, uint32_t* cigar, uint64_t cap_cigar,
                       uint32_t* bases2_out, uint32_t* nmask_out);
/* The same with the records going straight into a batch of the locus (as lcty_reads_append of the mapped chunk would put them):
 * records, CIGAR words and re-oriented bases are copied device to device, only the offsets visit the host. The device buffers of the
 * mapping (arenas and kernel scratch: tens of GB for long reads on many alleles) stay with the context from chunk to chunk; the solver
 * stages release them before they size their workspace, and so does lcty_ctx_trim. One mapping call at a time per context. */
int32_t lcty_reads_map_append(lcty_reads* reads, const lcty_reads_host* chunk, const lcty_map_params* params);

/* ---- solver stages (src/solvers/solve.rs:789-850, src/solvers/stoch.rs, src/model/assgn.rs) ----------------
 * The reference drives these stages from one Xoshiro256++ through rand ^0.10 adaptors that are not in its tree and
 * whose results already depend on --threads (solve.rs:1017, 1051). Here every (genotype, attempt) chain gets its
 * own 64-bit seed (lcty_chain_seeds draws them from seed_from_u64(master)); DESIGN.md §2 lists the adaptors. */
int32_t lcty_solver_default(lcty_solver* out, int32_t kind);                 /* Greedy::default / SimAnneal::default */
int32_t lcty_chain_seeds(uint64_t master_seed, uint64_t n, uint64_t* out);
/* One stage over genotypes[n_gt][ploidy] (the body of the stage loop, solve.rs:816-843): per genotype `attempts` x
 * (apply_tweak -> Solver::solve -> prior + likelihood), then mean_variance_or_nan. chain_seeds[n_gt*attempts];
 * liks_out (optional) [n_gt*attempts]. Device solver: ploidy <= 4. */
int32_t lcty_solve_stage(lcty_reads* reads, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy, const double* priors,
                         const lcty_solver* solver, uint32_t attempts, const uint64_t* chain_seeds,
                         double* lik_mean, double* lik_var, double* liks_out);

/* ---- `trait Solver` on the caller's own object (src/solvers/mod.rs:49-75) ------------------------------------------------
 * The reference calls a solver as `gt_alns.apply_tweak(rng, ..); stage.solver.solve(&gt_alns, rng)` (solve.rs:824-826): the
 * solver is handed a GenotypeAlignments (model/assgn.rs:16-36) the CALLER built and tweaked, and returns a ReadAssignment
 * (assgn.rs:171-186) borrowing it. lcty_gt_alns_view is that object as plain arrays, lcty_solve_given is `Solver::solve` on it:
 * ONE chain of `solver` (Greedy, stoch.rs:81-120; SimAnneal, 195-245; the exact model of highs.rs:38-134) on the device, over
 * exactly the locations, windows and window distributions given — nothing is re-derived from a read batch and no tweak is drawn.
 *   read_ixs[n_reads + 1]   assgn.rs:29-33; read pair i has the locations read_ixs[i] .. read_ixs[i + 1] (at least one, best first)
 *   ln_prob[n_alns]         ReadGtAlns::ln_prob (windows.rs:83-92)
 *   windows[2 * n_alns]     ReadGtAlns::windows after define_windows_determ / _random (windows.rs:112-136)
 *   window_gc / _weight[n_windows]   depth_distrs (assgn.rs:21, 140-150): WindowDistr::weight, 0 = WindowDistr::TRIVIAL
 *                           (distr_cache.rs:28-31); the GC bin selects the cached distribution of `locus` (DistrCache,
 *                           distr_cache.rs:61-92); windows 0 and 1 are the two trivial ones (assgn.rs:72-77)
 *   depth_contrib, aln_contrib   assgn.rs:24-25, 80-81
 *   wshifts[n_contigs + 1]  GenotypeWindows::wshifts (windows.rs:709-739), optional (n_contigs 0): only the exact solver looks
 *                           at it, for the order of its search
 * rng_state: the four words of the caller's XoshiroRng (xoshiro256++, ext/rand.rs:3). With non-trivial reads the call draws ONE
 * next_u64 from it — the chain's seed, as lcty_solve_stage takes one seed per chain; the chain is then that of lcty_solve_stage for
 * this seed (DESIGN.md §2 lists the adaptors) — and leaves the state advanced by that draw. A genotype without non-trivial reads has
 * one assignment (Solver::solve, mod.rs:64-66): the generator is not touched (rng_state may be NULL).
 * Outputs: read_assgn[n_reads] (ReadAssignment::read_assgn), lik_parts = {aln_lik, depth_lik} (optional), *likelihood =
 * depth_contrib * depth_lik + aln_contrib * aln_lik (ReadAssignment::likelihood, assgn.rs:235-237; optional).
 * Re-entrant: any number of host threads may call it at once on one locus (the Solver contract: `&self` shared by the worker
 * threads, solve.rs:254-257, 1010-1017); every call in flight has its own stream and chain state. Limits of the device solver:
 * < 2^24 read pairs, <= 255 locations per read pair (LCTY_ERR_UNSUPPORTED beyond; the reference asserts <= 65 535, assgn.rs:58).
 * LCTY_ERR_SOLVER as `Error::Solver` (err.rs:14): the exact solver without a proof inside its node limit (highs.rs:113-116). */
typedef struct lcty_gt_alns_view {
    uint64_t n_reads;
    const uint64_t* read_ixs;
    const double*   ln_prob;
    const uint32_t* windows;
    uint32_t n_windows;
    uint32_t n_contigs;
    const uint8_t*  window_gc;
    const double*   window_weight;
    const uint32_t* wshifts;
    double depth_contrib, aln_contrib;
} lcty_gt_alns_view;
int32_t lcty_solve_given(lcty_locus* locus, const lcty_gt_alns_view* gt_alns, const lcty_solver* solver, uint64_t* rng_state,
                         uint16_t* read_assgn, double* lik_parts, double* likelihood);
/* The same call for a caller that holds the window distributions itself and no lcty_locus — which is what `Solver::solve` is
 * handed: the WindowDistr of every window carries its cached distribution (Arc<LinearCache<BayesCalc<..>>>, distr_cache.rs:17-25), at
 * most one per GC bin. `tables`: values[n_rows][width] = DiscretePmf::ln_pmf(depth) of row's distribution for depth 0 .. width - 1
 * (LinearCache::ln_pmf, lincache.rs:41-48: cached below 256, evaluated beyond); window_gc[w] then names the ROW of window w. width
 * must exceed what lcty_gt_alns_deepest reports for the object: the number of locations that name the fullest window with a
 * distribution (no assignment can make a window deeper). id != 0: a table the slot already holds under this id, width and row count
 * is not uploaded again (the rows of a locus do not change between the attempts of its genotypes). n_rows <= 128. */
typedef struct lcty_depth_tables {
    uint32_t n_rows, width;
    const double* values;
    uint64_t id;
} lcty_depth_tables;
int32_t lcty_gt_alns_deepest(const lcty_gt_alns_view* gt_alns, uint32_t* deepest);
int32_t lcty_solve_given_tables(lcty_ctx* ctx, const lcty_gt_alns_view* gt_alns, const lcty_depth_tables* tables, const lcty_solver* solver,
                                uint64_t* rng_state, uint16_t* read_assgn, double* lik_parts, double* likelihood);
/* XoshiroRng::seed_from_u64 (ext/rand.rs:3-22) into four words / next_u64 on them: for a caller whose generator keeps its state
 * private (rand_xoshiro without serde): `lcty_rng_seed_from_u64(rng.next_u64(), state)` starts a stream for the solver calls */
int32_t lcty_rng_seed_from_u64(uint64_t seed, uint64_t* state);
int32_t lcty_rng_next_u64(uint64_t* state, uint64_t* out);
/* Per-read assignment counts of ONE genotype over `attempts` chains — the "per-read posteriors" behind the output BAMs
 * (GenotypeAlignments::create_counts + ReadAssignment::update_counts, assgn.rs:94-96, 374-378; solve.rs:821-836;
 * model/bam.rs divides by `attempts`). read_off[n_good + 1]: first count of every good read pair, its possible
 * locations on the genotype in extend_read_gt_alns order (windows.rs:762-797); counts[read_off[n_good]] (u16 as in the
 * reference). counts == NULL: only read_off / *n_counts are produced. The chains are the ones lcty_solve_stage runs
 * for the same genotype, solver and seeds. */
int32_t lcty_assignment_counts(lcty_reads* reads, const uint16_t* genotype, uint32_t ploidy, const lcty_solver* solver,
                               uint32_t attempts, const uint64_t* chain_seeds, uint64_t* read_off, uint16_t* counts, uint64_t cap,
                               uint64_t* n_counts);
/* Genotyping::count_unexplained_reads (solve.rs:718-729): good read pairs whose best alignment on the alleles of the
 * called genotype is no better than "both mates unmapped" (+1e-8). */
int32_t lcty_count_unexplained(lcty_reads* reads, const uint16_t* genotype, uint32_t ploidy, uint32_t* out);
/* Genotyping::{find_weighted_dist, check_first_prob, check_num_of_reads} (solve.rs:621-675). genotypes[n][ploidy] and
 * ln_probs[n] as produced by lcty_produce_result (best first); dist: optional n_alleles x n_alleles matrix of contig distances
 * (symmetric, LCTY_NONE_U32 = unknown; `contig_distances`, solve.rs:974-976). distances_out[n] (LCTY_NONE_U32 = None),
 * *weighted_dist (NaN = None), *warnings = LCTY_WARN_* bits. Host only. */
#define LCTY_WARN_NO_PROBABLE_GENOTYPE 1u
#define LCTY_WARN_FEW_READS            2u
int32_t lcty_call_checks(const uint16_t* genotypes, uint64_t n, uint32_t ploidy, const double* ln_probs, uint32_t n_reads,
                         const uint32_t* dist, uint32_t n_alleles, uint32_t* distances_out, double* weighted_dist, uint32_t* warnings);
/* ---- the whole genotyping of one locus: solve::solve (src/solvers/solve.rs:926-981) ---------------------------------
 * Scheme = list of stages (Stage, solve.rs:138-171; default "-S greedy:i=5k,a=1 -S anneal:i=20,a=20", 211-230). The call:
 * all genotypes of `ploidy` (generate_genotypes) with optional priors -> run_filter + truncate_ixs when the first stage
 * takes fewer genotypes than there are (or dont_skip) -> every stage (skipped when the survivors already fit the next
 * stage, 805-809) + discard_improbable_genotypes -> produce_result -> count_unexplained_reads / check_first_prob /
 * check_num_of_reads. Chain seeds of stage s: lcty_chain_seeds(master_seed + (s + 1) * 0x9e3779b97f4a7c15, n * attempts).
 * lik_mean / lik_var / attempts_out (optional, [G]): NaN / 0 for genotypes that were never solved. */
typedef struct lcty_stage {
    lcty_solver solver;
    uint64_t in_size;       /* genotypes this stage takes (Stage::in_size) */
    uint32_t attempts;      /* chains per genotype */
    uint32_t _pad0;
} lcty_stage;
#define LCTY_MAX_RESULT 50  /* MAX_GENOTYPES of produce_result (solve.rs:485) */
typedef struct lcty_call {
    uint64_t n_out;                       /* genotypes reported, best first */
    uint64_t ixs[LCTY_MAX_RESULT];        /* indices into generate_genotypes order */
    double   ln_probs[LCTY_MAX_RESULT];
    double   quality;                     /* Phred of "the call is wrong", capped at 1e9 */
    uint32_t unexpl_reads;                /* count_unexplained_reads of the call */
    uint32_t warnings;                    /* LCTY_WARN_* */
    uint64_t n_good;                      /* read pairs used */
    uint64_t kept_after_filter;           /* genotypes after run_filter (= all when it was skipped) */
} lcty_call;
int32_t lcty_stages_default(lcty_stage* stages /* [2] */, uint32_t* n_stages);
int32_t lcty_solve(lcty_reads* reads, uint32_t ploidy, const lcty_stage* stages, uint32_t n_stages, uint64_t master_seed,
                   const double* priors, lcty_call* out, double* lik_mean, double* lik_var, uint32_t* attempts_out);
/* The loop of `locityper genotype` over its loci (analyze_locus one after the other, command/genotype.rs:1331-1351) as a queue on one
 * GPU: for every entry lcty_score_reads + lcty_solve. Loci are independent, so the last stage of entry i (by default the annealing
 * attempts: a few hundred long serial chains on a few per cent of the device) runs on a second stream of the context, from a second
 * host thread, while entry i + 1 is greedily solved — and what comes before the chains of entry i + 2 (its scores, run_filter, the cut,
 * the location table) is issued on a third stream, by a third host thread, as soon as the last stage of entry i has ended, beside the
 * greedy chains of entry i + 1 (knob "queue_early_head" 0: on the main stream after those chains).
 * out[i] equals what lcty_solve gives for entry i alone.
 * All batches share one context; neighbours in the queue are different batches of different lcty_locus objects (a batch may come
 * again later in the queue: it is scored again). master_seeds[n_batches]; priors NULL or [n_batches] pointers (NULL = no priors). */
int32_t lcty_solve_queue(lcty_reads* const* batches, uint32_t n_batches, uint32_t ploidy, const lcty_stage* stages, uint32_t n_stages,
                         const uint64_t* master_seeds, const double* const* priors, lcty_call* out);
/* The same queue fed one batch at a time, for loci that are not all resident: `acquire(user, i)` is called right before position i is
 * scored and returns its batch (NULL: the queue ends with LCTY_ERR_INVALID_INPUT) — typically filled by another host thread, with
 * lcty_reads_append* on page-locked chunks (their copies have a stream of their own), while position i - 1 is being solved;
 * `release(user, i)` (may be NULL) when the last stage of position i is done and nothing of its batch is in use: lcty_reads_reset can
 * then bind the object to the locus of a later position. Position i is released before position i + 3 is acquired, so three batch
 * objects carry a queue of any length. `acquire(user, i)` for i > 0 comes from a thread of the library, while position i - 1 is in its
 * chains and once the last stage of position i - 2 has ended (the head of position i is made there: lcty_solve_queue), and may run at
 * the same time as a `release` on the caller's thread; with knob "queue_early_head" 0 both come from the caller's thread, `acquire(i)`
 * after the chains of position i - 1 and after `release(i - 2)`. On an error every position that was acquired and not yet released is
 * released before the call returns, with nothing of the queue left in flight on the device. The loading of locus i + 1 next to
 * analyze_locus of locus i (genotype.rs:1331-1351). */
typedef lcty_reads* (*lcty_queue_acquire_fn)(void* user, uint32_t position);
typedef void (*lcty_queue_release_fn)(void* user, uint32_t position);
int32_t lcty_solve_queue_fed(uint32_t n_loci, lcty_queue_acquire_fn acquire, lcty_queue_release_fn release, void* user, uint32_t ploidy,
                             const lcty_stage* stages, uint32_t n_stages, const uint64_t* master_seeds, const double* const* priors,
                             lcty_call* out);

/* Diagnostics of the last lcty_solve_stage on this batch: chains run, solver iterations (greedy iterations /
 * annealing moves) and accepted moves summed over the chains (stoch.rs has no counterpart; used by bench.py). */
int32_t lcty_solve_stats(const lcty_reads* reads, uint64_t* chains, uint64_t* iterations, uint64_t* accepted);
/* Predictions::discard_improbable_genotypes (solve.rs:425-480): ixs in/out */
int32_t lcty_discard_improbable(const double* lik_mean, const double* lik_var, const uint32_t* attempts, uint64_t* ixs, uint64_t n,
                                double prob_thresh, uint64_t out_size, uint64_t threads, uint64_t* n_keep);
/* Predictions::produce_result (solve.rs:482-535): out arrays sized >= min(n, 50) */
int32_t lcty_produce_result(const double* lik_mean, const double* lik_var, const uint32_t* attempts, const uint64_t* ixs, uint64_t n,
                            double prob_thresh, uint64_t out_bams, uint64_t* out_ixs, double* out_ln_probs, uint64_t* n_out,
                            double* quality);

/* ---- measurement hooks (bench.py) -----------------------------------------
 * HIP-event timing of the launches issued between begin/end on the context's
 * stream; kernel ids LCTY_K_*.                                                 */
#define LCTY_K_SCORE     0
#define LCTY_K_PREFILTER 1
#define LCTY_K_SOLVE     2   /* greedy_loop_kernel: the Greedy chains */
#define LCTY_K_SOLVE_INIT  3 /* solve_init_tile_kernel / solve_init_kernel: apply_tweak + ReadAssignment::try_new of every chain of a greedy (or exact) stage, its records */
#define LCTY_K_SOLVE_TABLE 4 /* build_loc_table_kernel: allele-major location table of a scored batch */
#define LCTY_K_TRANSFER  5   /* transfer_kernel: alignment recovery */
#define LCTY_K_RECRUIT   6   /* recruit_kernel: minimizer read recruitment */
#define LCTY_K_ANNEAL    7   /* anneal_loop_kernel: the SimAnneal chains */
#define LCTY_K_MAP       8   /* map_kernel: candidate generation on the basis alleles */
#define LCTY_K_SOLVE_INIT_ANNEAL 9 /* the same for the chains of an annealing stage (a few hundred: a launch of its own size, timed apart) */
#define LCTY_K_COUNT     10
/* Timing is opt-in: nothing is recorded before the first lcty_timing_reset on a context (a production run that never reads
 * timings creates no events); afterwards every launch is bracketed by two events, at most 256 pairs per kernel id kept. */
int32_t lcty_timing_reset(lcty_ctx* ctx);
int32_t lcty_timing_get(lcty_ctx* ctx, int32_t kernel, uint64_t* launches, double* total_ms);

/* ---- file formats at the edges of the path (host code, no device) --------------------------------------------
 * lcty_io_read_file: the whole file with its container removed, chosen by the extension as ext::sys::open does: .gz / .bgz / .bam
 *   gzip members (zlib), .lz4 LZ4 frames, .br brotli streams one after the other (src/ext/sys/brotli.rs:18-86; needs the system's
 *   libbrotlidec.so.1, LCTY_ERR_UNSUPPORTED without it), anything else as it is. *data is released with lcty_io_free.
 *   `kmers.bin.br` / `.lz4` (command/paths.rs:4-5) -> lcty_io_read_file -> lcty_kmer_counts_parse.
 * lcty_io_write_gz: ext::sys::create_gzip + write.
 * lcty_io_write_br: the brotli writer behind `sol.csv.br`, `reads.csv.br` and the other --debug tables (solvers/solve.rs:937-938,
 *   model/locs.rs:1062-1065): a brotli stream (RFC 7932) of `data` — compressed at `quality` 0..11 by the system's libbrotlienc.so.1
 *   (looked up at run time), or, without that library or with quality < 0, STORED in uncompressed meta-blocks (a valid stream any
 *   brotli reader takes, the reference's included; *stored = 1 then).
 * lcty_bg_from_json: BgDistr::load (src/bg/mod.rs:159-177) on the text of PREPROC/distr.gz: seq_info (349-364), insert_distr
 *   ({} = single-end, bg/insertsz.rs:195-208), error_profile (bg/err_prof.rs:321-329), bg_depth (bg/depth.rs:400-412; required, as
 *   `locityper genotype` requires it); edit thresholds = EditThresh::default_for (err_prof.rs:394-399). Missing keys / wrong types
 *   -> LCTY_ERR_INVALID_DATA (Error::JsonLoad).
 * lcty_res_to_json: Genotyping::to_json (src/solvers/solve.rs:732-773) in the layout of write_pretty(.., 4) (genotype.rs:1256):
 *   total_reads, quality, [dist_type, weight_dist], unexpl_reads, genotype, options[{genotype, lik_mean, lik_sd, prob, log10_prob,
 *   [dist_to_primary]}], [warnings]; lik_sd is the log10-scaled VARIANCE, as upstream (solve.rs:755). genotypes[n_out][ploidy],
 *   lik_mean / lik_var[n_out] in the order of call->ixs. Two calls: out = NULL sizes it (*needed includes the final 0). */
int32_t lcty_io_read_file(const char* path, uint8_t** data, uint64_t* len);
void    lcty_io_free(void* p);
int32_t lcty_io_write_gz(const char* path, const uint8_t* data, uint64_t len);
int32_t lcty_io_write_br(const char* path, const uint8_t* data, uint64_t len, int32_t quality, int32_t* stored);
int32_t lcty_bg_from_json(const char* json, uint64_t len, lcty_bg* bg, double* read_len);
int32_t lcty_res_to_json(const lcty_call* call, const uint16_t* genotypes, uint32_t ploidy, const char* const* names, uint32_t n_alleles,
                         const double* lik_mean, const double* lik_var, const uint32_t* distances, int32_t true_edit_distances,
                         double weighted_dist, char* out, uint64_t cap, uint64_t* needed);
/* write_bam (src/model/bam.rs:356-413): the alignments of the batch's read pairs to the contigs of ONE genotype as a coordinate-sorted
 * BAM file (path) and its BAI index (path + ".bai"). For every used read pair (status GOOD) the locations of the pair on the genotype
 * (extend_read_gt_alns, model/windows.rs:762-797) are folded by (alignment of mate 1, alignment of mate 2) with their assignment counts
 * (count_alignments, bam.rs:144-176) — read_off / counts exactly as lcty_assignment_counts returned them for THIS genotype and
 * `attempts` —, one record (pair) per fold: MAPQ and pr from the counts (count_to_prob, 56-67), the fold with most counts primary,
 * the others secondary; read pairs with few unique k-mers (status FEW_KMERS) follow with us:F (268-298, 328-353). Tags NM, il, al, uk,
 * pr, us as bam.rs:123-141; flags, mate fields and insert sizes as connect_pair / calc_insert_size (69-86, 178-221).
 *   table        the caller's copy of what was appended to the batch (records, CIGARs, packed sequences)
 *   name_off     [n_pairs + 1] into names; qual_off [2 * n_pairs + 1] into quals (NULL: qualities 255), both as in the primary
 *                record of each mate (lcty_bam_table_view hands them out for an aln.bam)
 * Not after lcty_recover_alignments (the transferred alignments are not in the caller's table) and not for counted batches:
 * LCTY_ERR_UNSUPPORTED. *n_records (may be NULL): records written. */
int32_t lcty_write_bam(const char* path, lcty_reads* reads, const lcty_reads_host* table, const uint64_t* name_off, const char* names,
                       const uint64_t* qual_off, const uint8_t* quals, const char* const* allele_names, const uint16_t* genotype,
                       uint32_t ploidy, uint16_t attempts, const uint64_t* read_off, const uint16_t* counts, uint64_t* n_records);

/* OUT/loci/<locus>/aln.bam -> the flat table of lcty_reads_append, grouped as AllAlignments::load walks the records
 * (src/model/locs.rs:405-461, 502-567, 1116-1150): a group = a primary record + the non-primary records behind it; the first group
 * of a read is its first end, with paired != 0 the next group (same name, else LCTY_ERR_INVALID_DATA as ReadData::set_name,
 * 147-155) its second. Reference names map to allele indices through names[n_alleles] (construct_tid_to_contig_map, 388-401).
 * lcty_bam_table_view: the table (valid while the handle lives), read names (name_off[n_pairs + 1] into name_blob) and the number
 * of reference sequences of the header (fewer than alleles: set lcty_params.strict_subset, locs.rs:486). */
typedef struct lcty_bam_table lcty_bam_table;
int32_t lcty_bam_read(const char* path, const char* const* names, uint32_t n_alleles, int32_t paired, lcty_bam_table** out);
int32_t lcty_bam_table_view(const lcty_bam_table* t, lcty_reads_host* view, const uint64_t** name_off, const char** name_blob, uint32_t* n_refs);
void    lcty_bam_table_free(lcty_bam_table* t);
/* DB/loci/<locus>/haplotypes.fa[.gz] (ContigSet::load, src/seq/contigs.rs:295-306): names up to the first blank (0-separated in
 * `names`), upper-cased sequences concatenated, seq_off[n_seqs + 1]. Called twice: with names = seqs = seq_off = NULL it returns the sizes. */
int32_t lcty_fasta_read(const char* path, uint32_t* n_seqs, char* names, uint64_t* names_len, uint8_t* seqs, uint64_t* seqs_len, uint64_t* seq_off);
/* DB/loci/<locus>/haplotypes.paf[.gz|.br|.lz4] as process_paf reads it (src/command/genotype.rs:1131-1160; PafFile::next and
 * PafEntry::parse, src/seq/paf.rs:31-56, 103-144): the entries lcty_locus_set_hap_alns takes, in file order. names[n_alleles]: the
 * contig names of the locus (their order gives the ids). Left out, as there: empty and '#' lines, lines naming a contig the locus
 * does not have, self-alignments, entries without a cg:Z: tag, entries that do not cover both sequences on the forward strand
 * (HapAlns::add, src/seq/transfer.rs:48-52). Malformed lines are errors, as there. Called twice: id1 = NULL sizes it
 * (*n_entries, *n_cigar); with buffers, *n_entries / *n_cigar carry their capacities in. cigar_off[n_entries + 1]. */
int32_t lcty_paf_read(const char* path, const char* const* names, uint32_t n_alleles, uint64_t* n_entries, uint32_t* id1, uint32_t* id2,
                      uint32_t* n_matches, uint32_t* aln_len, uint64_t* cigar_off, uint32_t* cigar, uint64_t* n_cigar,
                      uint32_t* dist /* NULL or [n_alleles x n_alleles]: contig_distances (genotype.rs:1139-1150), the edit distance
                                        aln_len - n_matches of every entry with a CIGAR between two different contigs of the locus,
                                        symmetric, LCTY_NONE_U32 where the file has none: the `dist` of lcty_call_checks, "edit" */);
/* DB/loci/<locus>/distances.bin (load_divergences_and_convert, src/seq/minim_div.rs:126-149; read when there is no PAF,
 * genotype.rs:1229-1237): u8 k, u8 w, varint n (= n_alleles or LCTY_ERR_INVALID_DATA), then the non-shared minimizers of the pairs
 * i < j row by row. dist[n_alleles x n_alleles]: symmetric, LCTY_NONE_U32 on the diagonal: the `dist` of lcty_call_checks, "minim-div". */
int32_t lcty_distances_parse(const uint8_t* buf, uint64_t len, uint32_t n_alleles, uint32_t* k, uint32_t* w, uint32_t* dist);

#ifdef __cplusplus
}
#endif
#endif /* LOCITYPER_HIP_H */
