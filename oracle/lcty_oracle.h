/*
 * lcty_oracle.h — CPU restatement of the Locityper scoring / prefilter / solver path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under locityper_amd/ or include/ may link,
 * import or execute this; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and only as the checker / timed baseline.
 *
 * PARITY UNPINNED: the reference (tprodanov/locityper v1.7.2, Rust) ships no tests,
 * golden vectors or fixtures for this path and cannot be built in this image
 * (no cargo/rustc, un-vendored crates, WFA2-lib not vendored). Each function
 * below restates the cited reference lines; distribution math is cross-checked
 * against scipy fixtures (tests/golden/), everything else against hand-derived
 * known answers.
 */
#ifndef LCTY_ORACLE_H
#define LCTY_ORACLE_H

#include <stdint.h>
#include <stddef.h>
#include "../include/locityper_hip.h"   /* plain-data structs only */

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned __int128 orc_u128;

/* ---- math ---------------------------------------------------------------- */
/* statrs 0.19 function::gamma::ln_gamma (Lanczos g=10.900511, n=11); call sites
 * src/math/distr/nbinom.rs:40,130 */
double orc_ln_gamma(double x);
double orc_ln_beta(double a, double b);                 /* statrs ln_beta; betabinom.rs:26,43,60 */
double orc_beta_reg(double a, double b, double x);      /* statrs beta_reg; nbinom.rs:146,150 */
double orc_ln_add(double a, double b);                  /* Ln::add  src/math/mod.rs:29-35 */
double orc_ln_sum_init(const double* v, size_t n, double init); /* Ln::map_sum_init src/math/mod.rs:80-94 */
double orc_ln_sum(const double* v, size_t n);           /* Ln::map_sum src/math/mod.rs:62-76 */

typedef struct { double n, p, lnq, lnpmf_const; } orc_nbinom;
orc_nbinom orc_nbinom_new(double n, double p);          /* nbinom.rs:35-42 */
double   orc_nbinom_ln_pmf(const orc_nbinom* d, uint32_t k);   /* nbinom.rs:128-131 */
uint32_t orc_nbinom_mode(const orc_nbinom* d);          /* nbinom.rs:78-80 */
double   orc_nbinom_cdf(const orc_nbinom* d, uint32_t k);      /* nbinom.rs:145-147 */
double   orc_nbinom_quantile(const orc_nbinom* d, double q);   /* math/distr/mod.rs:38-75 */
size_t   orc_insert_cache_size(const orc_nbinom* d);    /* bg/insertsz.rs:39-42 */
/* BetaBinomial::inv_cdf2 math/distr/betabinom.rs:74-102 */
void     orc_betabinom_inv_cdf2(double alpha, double beta, uint32_t n, double cdf1, double cdf2,
                                uint32_t* k1, uint32_t* k2);
/* EditDistCache::get_anew bg/err_prof.rs:434-443 */
void     orc_edit_thresholds(const lcty_bg* bg, uint32_t read_len, uint32_t* good, uint32_t* passable);
/* BayesCalc::ln_pmf with DistrCache::new's distributions, distr_cache.rs:61-75, bayes.rs:27-35 */
double   orc_depth_ln_pmf(const lcty_bg* bg, const lcty_params* prm, uint32_t gc, uint32_t depth);
/* one-sided unpaired t-test (math/mod.rs:180-220) incl. statrs StudentsT::cdf */
double   orc_students_t_cdf(double freedom, double x);
double   orc_t_test(double mean1, double var1, double mean2, double var2, double n);
double   orc_t_test_diffsizes(double mean1, double var1, double mean2, double var2, double n1, double n2);

/* model::Params::default + genotype.rs:1282-1296 auto fields */
void     orc_params_default(lcty_params* p);
int      orc_params_resolve(lcty_params* p, const lcty_bg* bg);

/* ---- k-mers --------------------------------------------------------------- */
/* kmers::kmers::<u128,_,CANONICAL> (seq/kmers.rs:163-202). out has n+1-k entries
 * (0 if n < k); UNDEF = all ones. Returns the number written. */
size_t orc_kmers_u128(const uint8_t* seq, size_t n, uint32_t k, int canonical, orc_u128* out);
size_t orc_kmers_u32(const uint8_t* seq, size_t n, uint32_t k, int canonical, uint32_t* out);
/* linguistic_complexity numerators (seq/compl.rs:115-140): out[n-w+1] distinct counts */
size_t orc_complexity_counts(const uint8_t* seq, size_t n, uint32_t k, uint32_t w, uint16_t* out);

/* ---- locus ----------------------------------------------------------------- */
typedef struct orc_locus orc_locus;
orc_locus* orc_locus_new(uint32_t n_alleles, const uint8_t* seqs, const uint64_t* seq_off,
                         const uint16_t* offtarget, const uint64_t* cnt_off, uint32_t k,
                         const lcty_bg* bg, const lcty_params* params);
void     orc_locus_free(orc_locus* l);
uint64_t orc_locus_n_unique_kmers(const orc_locus* l);
/* same products as lcty_locus_contig_info */
int      orc_locus_contig_info(const orc_locus* l, uint32_t allele, uint8_t* gc, uint32_t* uniq_cnt,
                               uint16_t* compl_cnt, uint32_t* n_windows, uint32_t* reg_start);
double   orc_locus_insert_lnprob(const orc_locus* l, uint32_t sz);  /* InsertDistr::ln_prob */
double   orc_locus_insert_penalty(const orc_locus* l);              /* InsertDistr::insert_penalty */
const lcty_params* orc_locus_params(const orc_locus* l);

/* ---- AllAlignments::load (locs.rs:1085-1185 + 1237-1288, no alignment recovery) ---- */
typedef struct orc_alns orc_alns;
/* returns NULL and sets *err (LCTY_ERR_*) on invalid data (the reference panics / errors) */
orc_alns* orc_load(const orc_locus* l, const lcty_reads_host* in, int* err);
void      orc_alns_free(orc_alns* a);
/* orc_load / orc_load_recover (hap != NULL) with the reference's thread structure: the BAM loop (read_next_alns, in_bounds,
 * calculate_read_weight, locs.rs:1116-1150) on the calling thread, recover_and_group_alignments on `threads` workers, the k-th
 * read that passes the loop on worker k % threads (1149, 1157-1174). Same results as orc_load. seconds[2] (optional): wall time
 * of the serial and of the grouping phase. The CPU baseline of bench.py. */
struct orc_hap_alns;
orc_alns* orc_load_mt(const orc_locus* l, const lcty_reads_host* in, const struct orc_hap_alns* hap, uint32_t threads, double* seconds, int* err);
/* test hook: orc_alns from arrays (status, weight, unmapped_prob [n_pairs], pa_off [n_pairs+1], pa) */
orc_alns* orc_alns_from_arrays(uint64_t n_pairs, uint32_t n_alleles, const uint8_t* status, const double* weight,
                               const double* unmapped_prob, const uint64_t* pa_off, const lcty_pair_aln* pa);
uint64_t  orc_alns_n_pairs(const orc_alns* a);
uint64_t  orc_alns_n_good(const orc_alns* a);
void      orc_alns_status(const orc_alns* a, uint8_t* status, double* weight, double* unmapped_prob,
                          uint16_t* uniq_kmers);
/* CSR over all input pairs, see lcty_reads_get_pair_alns */
uint64_t  orc_alns_pair_alns(const orc_alns* a, uint64_t* off, lcty_pair_aln* out, uint64_t cap);
/* best_aln_matrix (locs.rs:1203-1212): out[a*n_good + j] */
void      orc_best_aln_matrix(const orc_alns* a, double* out);

/* ---- genotypes / prefilter -------------------------------------------------- */
uint64_t orc_count_genotypes(uint32_t n_alleles, uint32_t ploidy);          /* ext/vec.rs:285-296 */
uint64_t orc_generate_genotypes(uint32_t n_alleles, uint32_t ploidy, uint16_t* out); /* ext/vec.rs:298-339 */
/* run_filter scores (solvers/solve.rs:101-119), matrix = [A][n_good] */
void     orc_run_filter(const double* matrix, uint32_t n_alleles, uint64_t n_good,
                        const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy,
                        const double* priors, double* scores);
/* truncate_ixs (solvers/solve.rs:52-84), ties broken by index ascending */
uint64_t orc_truncate(const double* scores, uint64_t* ixs, uint64_t n, double filt_diff,
                      uint64_t min_size, uint64_t threads);


/* ======================================================================================== */
/* solver stages                                                 */
/* ======================================================================================== */

/* Xoshiro256PlusPlus::seed_from_u64 (ext/rand.rs:3-22; SplitMix64 expansion), next_u64, jump
 * (solve.rs:1017), long_jump (genotype.rs:1345): public-domain reference constants. */
typedef struct { uint64_t s[4]; } orc_rng;
void     orc_rng_seed(orc_rng* r, uint64_t seed);
uint64_t orc_rng_next(orc_rng* r);
void     orc_rng_jump(orc_rng* r);
void     orc_rng_long_jump(orc_rng* r);
/* Sampling adaptors. rand ^0.10's own adaptors are not in the reference tree, so these are OUR
 * documented definitions, shared by oracle and GPU:
 *   below(n)  = high 64 bits of next_u64() * n          (stands in for random_range(0..n))
 *   f64()     = (next_u64() >> 11) * 2^-53              (random::<f64>())
 *   counter(key, i) = SplitMix64 finaliser of key + (i+1) * 0x9e3779b97f4a7c15  (order-free draws of apply_tweak)
 *   sample(k of n)  = key = next_u64(); picks = high 64 bits of counter(key, 0), counter(key, 1), ... times n, repeats skipped
 *                     (IndexedRandom::sample of the greedy loop: k distinct indices, one generator draw per iteration) */
uint64_t orc_rng_below(orc_rng* r, uint64_t n);
double   orc_rng_f64(orc_rng* r);
uint64_t orc_counter_u64(uint64_t key, uint64_t i);

/* WeightCalculator::get (model/windows.rs:163-177) */
double   orc_weight_calc(double breakpoint, double power, double x);
/* ContigInfo::neighb_info weight for a window starting at `wstart` (windows.rs:439-445); *gc = NeighbInfo::gc_content */
double   orc_window_weight(const orc_locus* l, uint32_t allele, uint32_t wstart, uint32_t* gc);
/* bulk forms for the pin tests: every per-position window weight (the oracle's own, not the injected ones); BayesCalc::ln_pmf
 * for every GC bin and depth lo..hi-1 */
void     orc_locus_window_weights(const orc_locus* l, double* out);
void     orc_depth_table(const lcty_bg* bg, const lcty_params* prm, uint32_t lo, uint32_t hi, double* out);
/* WindowDistr::ln_prob of the depth LUT / direct evaluation (distr_cache.rs:34-39, lincache.rs:41-48) */
double   orc_depth_ln_prob(const orc_locus* l, uint32_t gc, double weight, uint32_t depth);

/* GenotypeAlignments (model/assgn.rs:16-169) + GenotypeWindows (windows.rs:709-806) */
typedef struct orc_gt_alns orc_gt_alns;
orc_gt_alns* orc_gt_alns_new(const orc_locus* l, const orc_alns* a, const uint16_t* ids, uint32_t ploidy);
void     orc_gt_alns_free(orc_gt_alns* g);
uint64_t orc_gt_alns_n_reads(const orc_gt_alns* g);
uint64_t orc_gt_alns_n_alns(const orc_gt_alns* g);
uint32_t orc_gt_alns_n_windows(const orc_gt_alns* g);
uint64_t orc_gt_alns_n_nontrivial(const orc_gt_alns* g);
/* any pointer may be NULL. contig_ix 0xFF = both mates unmapped; mids LCTY_NONE_U32 = None; windows[2*i..] */
void     orc_gt_alns_get(const orc_gt_alns* g, uint64_t* read_ixs, double* ln_prob, uint8_t* contig_ix,
                         uint32_t* mid1, uint32_t* mid2, uint32_t* windows, uint64_t* non_trivial);
/* apply_tweak (assgn.rs:127-151): location t of read rp draws counter(key, rp << 16 | t); window w of the genotype
 * draws counter(key ^ 0xD1B54A32D192ED03, w); a random initial assignment of read rp draws
 * counter(key ^ 0x8CB92BA72F3D8DD7, rp) */
void     orc_gt_alns_apply_tweak(orc_gt_alns* g, uint64_t key);
/* per window after apply_tweak: gc and weight (weight 0.0 = WindowDistr::TRIVIAL) */
void     orc_gt_alns_window_distr(const orc_gt_alns* g, uint8_t* gc, double* weight);
double   orc_gt_alns_max_aln_lik(const orc_gt_alns* g);

void     orc_solver_default(lcty_solver* s, int32_t kind);
/* Solver::solve (solvers/mod.rs:57-72) -> ReadAssignment::likelihood() (assgn.rs:235-237);
 * assgn_out[n_reads] = chosen location per read; lik_parts = {aln_lik, depth_lik} */
double   orc_solve(const orc_gt_alns* g, const lcty_solver* s, uint64_t seed, uint16_t* assgn_out, double* lik_parts);
/* load_explicit_weights (model/windows.rs:257-317) without the text parsing: the lines of the BED file in file order as
 * (allele, start, end, value); an allele index >= n_alleles is an unknown contig (line skipped, 269-272). Errors as upstream
 * (value outside [0, 1]; an allele not covered from its start, missing, or of different length: ParsingError ->
 * LCTY_ERR_INVALID_DATA; an interval beyond the end of its allele: LCTY_ERR_INVALID_INPUT, interv.rs:112-116).
 * From then on window weights carry the window's average (409-413, 443) and read pairs their explicit_read_weight (683-693). */
int      orc_locus_set_explicit_weights(orc_locus* l, uint32_t n, const uint32_t* allele, const uint32_t* start, const uint32_t* end,
                                        const double* value);
/* test hook: depth_lut[101*256] and/or win_weight (per position, alleles concatenated) replace the oracle's own tables */
void     orc_locus_inject_tables(orc_locus* l, const double* depth_lut, const double* win_weight);
void     orc_locus_inject_depth_table(orc_locus* l, uint32_t width, const double* table);
/* likelihood of an explicit assignment (recalc_likelihood, assgn.rs:346-354) */
double   orc_assignment_likelihood(const orc_gt_alns* g, const uint16_t* assgn, double* lik_parts);

/* One stage of solve_single_thread (solve.rs:816-843) for genotypes[n_gt][ploidy]: chain (g, attempt) is
 * driven by chain_seeds[g*attempts + attempt] (tweak key = seed, solver rng = seed_from_u64(seed)).
 * Outputs lik_mean / lik_var (mean_variance_or_nan, ext/vec.rs:109-116); counts (optional) receives, per
 * genotype, n_alns(g) u16 values appended; counts_off[n_gt+1]. */
void     orc_solve_stage(const orc_locus* l, const orc_alns* a, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy,
                         const double* priors, const lcty_solver* s, uint32_t attempts, const uint64_t* chain_seeds,
                         double* lik_mean, double* lik_var, double* liks_out);

/* orc_solve_stage on `threads` worker threads, the stage's genotypes dealt in contiguous runs as MainWorker::run does
 * (solve.rs:1047-1062); same results for any number of threads. The CPU baseline of bench.py. */
void     orc_solve_stage_mt(const orc_locus* l, const orc_alns* a, const uint16_t* genotypes, uint64_t n_gt, uint32_t ploidy,
                            const double* priors, const lcty_solver* s, uint32_t attempts, const uint64_t* chain_seeds,
                            double* lik_mean, double* lik_var, double* liks_out, uint32_t threads);

/* per-read assignment counts of one genotype over its attempts (assgn.rs:94-96, 374-378; solve.rs:821-836) */
uint64_t orc_assignment_counts(const orc_locus* l, const orc_alns* a, const uint16_t* ids, uint32_t ploidy, const lcty_solver* s,
                               uint32_t attempts, const uint64_t* chain_seeds, uint64_t* read_ixs_out, uint16_t* counts_out);

/* ---- alignment recovery (lcty_oracle_transfer.c). Haplotype-to-haplotype alignments as HapAlns::add
 * takes them (transfer.rs:41-62): id1 = query contig, id2 = target contig, a full forward alignment as raw BAM CIGAR words
 * (=, X, I, D), its number of matches and its length (paf.rs:191-208). transfer_fails / max_div: genotype.rs defaults. */
typedef struct orc_hap_alns orc_hap_alns;
orc_hap_alns* orc_hap_alns_new(uint32_t n_contigs, uint32_t transfer_fails, double max_div);
void orc_hap_alns_free(orc_hap_alns* h);
void orc_hap_alns_add(orc_hap_alns* h, uint32_t id1, uint32_t id2, const uint32_t* cigar, uint32_t n_cigar, uint32_t n_matches,
                      uint32_t aln_len);
void orc_hap_alns_sort(orc_hap_alns* h);
/* AllAlignments::load with alignment recovery (locs.rs:1237-1288 with opt_hap_alns = Some) */
orc_alns* orc_load_recover(const orc_locus* l, const lcty_reads_host* in, const orc_hap_alns* hap, int* err);
/* test hook: one transfer (Cigar::transfer_read_alignment behind HapAlns): returns the new start, writes raw CIGAR words */
uint32_t orc_transfer_one(const orc_hap_alns* h, uint32_t source, uint32_t target, uint32_t source_start, const uint32_t* read_cigar,
                          uint32_t n_read_cigar, const uint8_t* read_seq, uint32_t read_len, const uint8_t* target_seq,
                          uint32_t target_len, uint32_t* out_cigar, uint32_t out_cap, uint32_t* n_out);

void orc_transfer_set_optimize(int on);     /* test switch, see lcty_oracle_transfer.c */
int orc_dp_align(const uint8_t* s1, uint32_t n, const uint8_t* s2, uint32_t m, int match_bonus, int mode, uint32_t* out_cigar, uint32_t out_cap,
                 uint32_t* n_out);

/* compare_two_likelihoods (solve.rs:319-336) */
double   orc_compare_two_likelihoods(double mean1, double var1, uint32_t att1, double mean2, double var2, uint32_t att2);
/* discard_improbable_genotypes (solve.rs:425-480): ixs in/out, returns the new count */
uint64_t orc_discard_improbable(const double* lik_mean, const double* lik_var, const uint32_t* attempts, uint64_t* ixs,
                                uint64_t n, double prob_thresh, uint64_t out_size, uint64_t threads);
/* produce_result (solve.rs:482-535): out_ixs/out_ln_probs sized >= min(n, 50); returns the number of genotypes, *quality */
uint64_t orc_produce_result(const double* lik_mean, const double* lik_var, const uint32_t* attempts, const uint64_t* ixs,
                            uint64_t n, double prob_thresh, uint64_t out_bams, uint64_t* out_ixs, double* out_ln_probs,
                            double* quality);
/* find_weighted_dist / check_first_prob / check_num_of_reads (solve.rs:621-675) */
void orc_call_checks(const uint16_t* genotypes, uint64_t n, uint32_t ploidy, const double* ln_probs, uint32_t n_reads,
                     const uint32_t* dist, uint32_t n_alleles, uint32_t* distances_out, double* weighted_dist, uint32_t* warnings);
/* count_unexplained_reads (solve.rs:718-729) */
uint32_t orc_count_unexplained(const orc_alns* a, const uint16_t* ids, uint32_t ploidy);


/* ---- minimizer read recruitment (src/seq/recruit.rs, src/seq/kmers.rs:71-340, src/math/frac.rs) — lcty_oracle_recruit.c ---- */
uint64_t orc_fast_hash64(uint64_t x);                                                        /* kmers.rs:93-103 */
size_t   orc_canon_minimizers(const uint8_t* seq, size_t n, uint32_t k, uint32_t w, uint32_t* pos_out, uint64_t* hash_out,
                              uint8_t* fw_out, size_t cap);                                  /* kmers.rs:265-340 */
void     orc_fraction_approximate_u16(double x, uint16_t* num, uint16_t* den);               /* frac.rs:50-76 */
typedef struct orc_targets orc_targets;
orc_targets* orc_targets_new(uint8_t k, uint8_t w, double match_frac, uint32_t match_length, uint16_t thresh_kmer_count);   /* Params::new */
void     orc_targets_free(orc_targets* t);
void     orc_targets_params(const orc_targets* t, uint16_t* mf_num, uint16_t* mf_den, uint32_t* stretch_minims, uint32_t* stretch_score);
uint32_t orc_targets_add_locus(orc_targets* t, uint32_t n_alleles, const uint8_t* seqs, const uint64_t* seq_off, const uint16_t* counts,
                               const uint64_t* cnt_off, uint32_t base_k);                    /* TargetBuilder::add */
void     orc_targets_finalize(orc_targets* t);
size_t   orc_targets_n_entries(const orc_targets* t);
void     orc_targets_entry(const orc_targets* t, size_t i, uint64_t* minim, uint32_t* locus, uint8_t* direction, uint8_t* rare);
/* recruit_read_pair (seq2 != NULL) / recruit_short_read / recruit_long_read: loci in increasing order */
size_t   orc_recruit(const orc_targets* t, const uint8_t* seq1, size_t n1, const uint8_t* seq2, size_t n2, uint32_t* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
