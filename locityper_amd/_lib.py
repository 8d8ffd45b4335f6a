"""Loader of liblocityper_hip.so (the C-ABI product library). Fails loudly: there is no CPU fallback."""
import ctypes as C
import os

from .cdefs import Bg, Params, ReadsHost, PairAln, Solver, Stage, Call, GtAlnsView, DepthTables

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblocityper_hip.so")
DIAG_LIB_PATH = os.path.join(_HERE, "liblocityper_hip_diag.so")     # the developer build (make -C locityper_amd/csrc DIAG=1): traces, in-kernel timing
_lib = None


def use_diag_build():
    """Developer probes (scripts/): load the -DLCTY_DIAG build instead of the product library; before the first call into it."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("the library is loaded already")
    if not os.path.exists(DIAG_LIB_PATH):
        raise RuntimeError("no developer build: make -C locityper_amd/csrc DIAG=1")
    LIB_PATH = DIAG_LIB_PATH

I32, U32, U64, D, VP = C.c_int32, C.c_uint32, C.c_uint64, C.c_double, C.c_void_p
P = C.POINTER

# name -> (restype, argtypes): every symbol declared in include/locityper_hip.h
SIGNATURES = {
    "lcty_last_error": (C.c_char_p, []),
    "lcty_version": (C.c_char_p, []),
    "lcty_device_count": (I32, []),
    "lcty_ctx_create": (I32, [I32, P(VP)]),
    "lcty_ctx_destroy": (None, [VP]),
    "lcty_ctx_synchronize": (I32, [VP]),
    "lcty_params_default": (None, [P(Params)]),
    "lcty_params_resolve": (I32, [P(Params), P(Bg)]),
    "lcty_locus_create": (I32, [VP, U32, VP, VP, VP, VP, U32, P(Bg), P(Params), P(VP)]),
    "lcty_locus_destroy": (None, [VP]),
    "lcty_locus_n_unique_kmers": (I32, [VP, P(U64)]),
    "lcty_locus_contig_info": (I32, [VP, U32, VP, VP, VP, P(U32), P(U32)]),
    "lcty_locus_edit_thresholds": (I32, [VP, U32, P(U32), P(U32)]),
    "lcty_locus_insert_lnprob": (I32, [VP, U32, VP, VP, P(D)]),
    "lcty_locus_depth_lut": (I32, [VP, VP]),
    "lcty_reads_create": (I32, [VP, U64, U64, U64, U64, P(VP)]),
    "lcty_reads_create_streaming": (I32, [VP, U64, U64, U64, U64, U64, U64, P(VP)]),
    "lcty_reads_append_counted": (I32, [VP, VP, VP]),
    "lcty_reads_append": (I32, [VP, P(ReadsHost)]),
    "lcty_reads_destroy": (None, [VP]),
    "lcty_reads_n_pairs": (I32, [VP, P(U64)]),
    "lcty_score_reads": (I32, [VP]),
    "lcty_reads_get_status": (I32, [VP, VP, VP, VP, VP]),
    "lcty_reads_n_good": (I32, [VP, P(U64)]),
    "lcty_best_aln_matrix": (I32, [VP, VP]),
    "lcty_reads_get_pair_alns": (I32, [VP, VP, VP, U64]),
    "lcty_reads_get_records": (I32, [VP, VP, VP, U64, VP, VP, U64]),
    "lcty_prefilter": (I32, [VP, VP, U64, U32, VP, VP]),
    "lcty_prefilter_async": (I32, [VP, U32]),
    "lcty_prefilter_scores": (I32, [VP, VP, U64]),
    "lcty_truncate": (I32, [VP, VP, U64, D, U64, U64, P(U64)]),
    "lcty_prefilter_truncate": (I32, [VP, D, U64, U64, VP, U64, P(U64)]),
    "lcty_prefilter_add_priors": (I32, [VP, VP, U64]),
    "lcty_count_genotypes": (U64, [U32, U32]),
    "lcty_generate_genotypes": (I32, [U32, U32, VP, U64]),
    "lcty_locus_window_weights": (I32, [VP, VP]),
    "lcty_kmer_counts_parse": (I32, [VP, U64, P(U32), P(U32), VP, U64, VP, U64, P(U64)]),
    "lcty_locus_set_explicit_weights": (I32, [VP, U32, VP, VP, VP, VP]),
    "lcty_locus_set_hap_alns": (I32, [VP, U32, VP, VP, VP, VP, VP, VP, U32, D]),
    "lcty_recover_alignments": (I32, [VP, P(U64)]),
    "lcty_recover_dp_cells": (I32, [VP, P(U64)]),
    "lcty_recover_stats": (I32, [VP, P(U64)]),
    "lcty_counts_to_posteriors": (I32, [VP, U64, C.c_uint16, VP, VP]),
    "lcty_comm_unique_id": (I32, [VP]),
    "lcty_comm_create": (I32, [VP, I32, I32, VP, P(VP)]),
    "lcty_comm_destroy": (None, [VP]),
    "lcty_comm_ranks": (I32, [VP, P(I32), P(I32)]),
    "lcty_prefilter_allreduce": (I32, [VP, VP]),
    "lcty_recruit_params_default": (I32, [VP, I32, I32]),
    "lcty_targets_create": (I32, [VP, VP, P(VP)]),
    "lcty_targets_destroy": (None, [VP]),
    "lcty_targets_add_locus": (I32, [VP, U32, VP, VP, VP, VP, U32, P(U32)]),
    "lcty_targets_finalize": (I32, [VP, P(U64)]),
    "lcty_recruit": (I32, [VP, VP, I32, U32, VP, VP]),
    "lcty_fastx_open": (I32, [C.c_char_p, C.c_char_p, I32, P(VP)]),
    "lcty_fastx_close": (None, [VP]),
    "lcty_fastx_is_paired": (I32, [VP, P(I32)]),
    "lcty_fastx_next": (I32, [VP, U64, P(ReadsHost), P(U64)]),
    "lcty_fastx_writers_open": (I32, [P(C.c_char_p), U32, P(VP)]),
    "lcty_fastx_write_recruited": (I32, [VP, VP, U32, VP, VP, P(U64)]),
    "lcty_fastx_writers_close": (I32, [VP]),
    "lcty_locus_depth_table": (I32, [VP, P(U32), VP]),
    "lcty_solver_default": (I32, [P(Solver), I32]),
    "lcty_chain_seeds": (I32, [U64, U64, VP]),
    "lcty_solve_stage": (I32, [VP, VP, U64, U32, VP, P(Solver), U32, VP, VP, VP, VP]),
    "lcty_solve_given": (I32, [VP, P(GtAlnsView), P(Solver), VP, VP, VP, P(D)]),
    "lcty_solve_given_tables": (I32, [VP, P(GtAlnsView), P(DepthTables), P(Solver), VP, VP, VP, P(D)]),
    "lcty_gt_alns_deepest": (I32, [P(GtAlnsView), P(U32)]),
    "lcty_rng_seed_from_u64": (I32, [U64, VP]),
    "lcty_rng_next_u64": (I32, [VP, P(U64)]),
    "lcty_solve_stage_sharded": (I32, [VP, VP, VP, U64, U32, VP, P(Solver), U32, VP, VP, VP, VP]),
    "lcty_solve_stage_read_sharded": (I32, [VP, VP, VP, U64, U32, VP, P(Solver), U32, VP, VP, VP, VP]),
    "lcty_solve_stage_from_shards": (I32, [VP, U32, VP, U64, U32, VP, P(Solver), U32, VP, VP, VP, VP]),
    "lcty_assignment_counts": (I32, [VP, VP, U32, P(Solver), U32, VP, VP, VP, U64, P(U64)]),
    "lcty_count_unexplained": (I32, [VP, VP, U32, P(U32)]),
    "lcty_call_checks": (I32, [VP, U64, U32, VP, U32, VP, U32, VP, P(D), P(U32)]),
    "lcty_stages_default": (I32, [P(Stage), P(U32)]),
    "lcty_solve": (I32, [VP, U32, P(Stage), U32, U64, VP, P(Call), VP, VP, VP]),
    "lcty_solve_queue": (I32, [VP, U32, U32, P(Stage), U32, VP, VP, VP]),
    "lcty_solve_queue_fed": (I32, [U32, VP, VP, VP, U32, P(Stage), U32, VP, VP, VP]),
    "lcty_reads_reset": (I32, [VP, VP]),
    "lcty_solve_stats": (I32, [VP, P(U64), P(U64), P(U64)]),
    "lcty_discard_improbable": (I32, [VP, VP, VP, VP, U64, D, U64, U64, P(U64)]),
    "lcty_produce_result": (I32, [VP, VP, VP, VP, U64, D, U64, VP, VP, P(U64), P(D)]),
    "lcty_ctx_set_knob": (I32, [VP, C.c_char_p, C.c_int64]),
    "lcty_ctx_set_path": (I32, [VP, C.c_char_p, C.c_char_p]),
    "lcty_map_params_default": (I32, [VP]),
    "lcty_map_params_default_long": (I32, [VP]),
    "lcty_locus_build_map_index": (I32, [VP, VP, U32, U32]),
    "lcty_map_reads": (I32, [VP, VP, VP, VP, VP, U64, VP, VP, U64, VP, VP]),
    "lcty_reads_map_append": (I32, [VP, VP, VP]),
    "lcty_ctx_trim": (I32, [VP]),
    "lcty_host_alloc": (I32, [VP, U64, P(VP)]),
    "lcty_host_free": (None, [VP]),
    "lcty_io_read_file": (I32, [C.c_char_p, P(VP), P(U64)]),
    "lcty_io_free": (None, [VP]),
    "lcty_io_write_gz": (I32, [C.c_char_p, VP, U64]),
    "lcty_io_write_br": (I32, [C.c_char_p, VP, U64, I32, P(I32)]),
    "lcty_bg_from_json": (I32, [C.c_char_p, U64, VP, P(D)]),
    "lcty_res_to_json": (I32, [VP, VP, U32, VP, U32, VP, VP, VP, I32, D, VP, U64, P(U64)]),
    "lcty_write_bam": (I32, [C.c_char_p, VP, VP, VP, VP, VP, VP, VP, VP, U32, C.c_uint16, VP, VP, P(U64)]),
    "lcty_bam_read": (I32, [C.c_char_p, VP, U32, I32, P(VP)]),
    "lcty_bam_table_view": (I32, [VP, VP, P(VP), P(VP), P(U32)]),
    "lcty_bam_table_free": (None, [VP]),
    "lcty_fasta_read": (I32, [C.c_char_p, P(U32), VP, P(U64), VP, P(U64), VP]),
    "lcty_paf_read": (I32, [C.c_char_p, P(C.c_char_p), U32, P(U64), VP, VP, VP, VP, VP, VP, P(U64), VP]),
    "lcty_distances_parse": (I32, [VP, U64, U32, P(U32), P(U32), VP]),
    "lcty_timing_reset": (I32, [VP]),
    "lcty_timing_get": (I32, [VP, I32, P(U64), P(D)]),
}


class LocityperError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[lcty error {code}] {msg}")
        self.code = code


def lib():
    """The loaded library; raises if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C locityper_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise LocityperError(rc, lib().lcty_last_error().decode())
