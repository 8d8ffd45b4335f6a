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
};

struct orc_alns {
    uint64_t n_pairs, n_good;
    uint32_t n_alleles;
    uint8_t* status; double* weight; double* unmapped_prob; uint16_t* uniq_kmers;
    uint64_t* pa_off; lcty_pair_aln* pa; size_t n_pa, cap_pa;
};

#endif
