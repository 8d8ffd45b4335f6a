/* lcty_oracle_internal.h — structs shared by the oracle's translation units. TEST INFRASTRUCTURE ONLY. */
#ifndef LCTY_ORACLE_INTERNAL_H
#define LCTY_ORACLE_INTERNAL_H
#include "lcty_oracle.h"

typedef struct {
    orc_u128* keys;
    uint8_t* used;
    size_t cap, len;
} u128set;

typedef struct {
    uint32_t len;
    uint32_t n_pos;        /* len - neighb + 1 */
    uint32_t n_windows, reg_start, reg_end;
    uint8_t*  gc;          /* NeighbInfo::gc_content */
    uint32_t* uniq_cnt;    /* numerator of uniq_kmer_frac */
    uint16_t* compl_cnt;   /* numerator of complexity */
} orc_contig_info;

struct orc_locus {
    uint32_t n_alleles, k;
    lcty_bg bg;
    lcty_params prm;
    const uint8_t** seqs;     /* borrowed copies */
    uint8_t* seq_store;
    uint64_t* seq_off;
    orc_contig_info* infos;
    u128set unique;
    double weight_mult, weight_interc;
    /* InsertDistr */
    orc_nbinom ins;
    double* ins_lut; size_t ins_lut_size; double ins_mode_prob;
    /* ContigInfo scalars */
    uint32_t left_padding, half_neighb;
    double uniq_mult, compl_mult;
    double* depth_lut;      /* [101][256] LinearCache<BayesCalc> values (distr_cache.rs:61-75) */
    double* win_weight_inj; uint64_t* ci_off_inj;   /* test hook: injected per-position window weights */
    double* depth_ext_inj; uint32_t depth_ext_width; /* test hook: injected BayesCalc::ln_pmf values [101][width] for depths beyond the LinearCache */
    /* ExplicitWeights per allele (model/windows.rs:196-250): len + 1 entries (value, running fixed-point sum) after finish() */
    int has_explicit;
    double** ew_val; uint64_t** ew_cum;
};
/* ExplicitWeights::{at,average} and ContigInfo::read_end_weight (windows.rs:226-238, 493-503) */
double orc_explicit_average(const orc_locus* l, uint32_t allele, uint32_t i, uint32_t j);
double orc_read_end_weight(const orc_locus* l, uint32_t allele, uint32_t middle /* LCTY_NONE_U32 = None */);

/* ---- alignment recovery (lcty_oracle_transfer.c) ---- */
typedef struct { uint32_t op, len; } orc_citem;                    /* CigarItem, BAM operation codes */
typedef struct { orc_citem* t; uint32_t n, cap, rlen, qlen; } orc_cigar;   /* Cigar — cigar.rs:203-208 */
void orc_cigar_init(orc_cigar* c);
void orc_cigar_free(orc_cigar* c);
void orc_cigar_clear(orc_cigar* c);
void orc_cigar_push_unchecked(orc_cigar* c, uint32_t op, uint32_t len);
void orc_cigar_push_checked(orc_cigar* c, uint32_t op, uint32_t len);
void orc_cigar_copy(orc_cigar* dst, const orc_cigar* src);
void orc_cigar_from_raw(orc_cigar* c, const uint32_t* raw, uint32_t n, int hard_to_soft);
uint32_t orc_cigar_to_raw(const orc_cigar* c, uint32_t* out);
uint32_t orc_transfer_read_alignment(const orc_cigar* cigar_jk, int dir_jk, uint32_t start_j, uint32_t off_cigar_ix, uint32_t off_qpos,
                                     uint32_t off_rpos, const orc_cigar* cigar_ij, const uint8_t* seq_i, uint32_t len_i,
                                     const uint8_t* seq_k, uint32_t len_k, orc_cigar* out);
uint32_t orc_hap_alns_n_best(const orc_hap_alns* h, uint32_t contig);
uint32_t orc_hap_alns_best(const orc_hap_alns* h, uint32_t contig, uint32_t i);
uint32_t orc_hap_alns_transfer_fails(const orc_hap_alns* h);
uint32_t orc_hap_alns_approx_pos(const orc_hap_alns* h, uint32_t source, uint32_t target, uint32_t source_start);
uint32_t orc_hap_alns_transfer(const orc_hap_alns* h, uint32_t source, uint32_t target, uint32_t source_start, const orc_cigar* source_cigar,
                               const uint8_t* read_seq, uint32_t read_len, const uint8_t* target_seq, uint32_t target_len, orc_cigar* out);

struct orc_alns {
    uint64_t n_pairs, n_good;
    uint32_t n_alleles;
    uint8_t* status; double* weight; double* unmapped_prob; uint16_t* uniq_kmers;
    uint64_t* pa_off; lcty_pair_aln* pa; size_t n_pa, cap_pa;
};

#endif
