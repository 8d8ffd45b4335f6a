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

#ifdef __cplusplus
}
#endif
#endif
