"""BASELINE.json configs[2] on the path the reference runs — 10-kb ONT reads with their alignments GIVEN (as minimap2 supplies them,
src/command/genotype.rs:990-1002), then AllAlignments::load (src/model/locs.rs:873-911: single-end grouping; unmapped_penalty 1e-100,
src/model/mod.rs:55-60) -> recover_and_group_alignments (locs.rs:1237-1288) -> run_filter -> the default scheme -> the call — through
the C ABI, streamed (bench_legs/long_reads.py::ont_whole_path is the harness the bench line uses). Needs an MI355X."""
import numpy as np
import pytest

from bench_legs import cpu as CPU
from bench_legs import long_reads as LR

pytestmark = pytest.mark.gpu


def check(out, n_reads):
    assert out["all_calls_equal_truth"], (out["called_genotype"], out["true_genotype"])
    assert out["chains_equal_oracle"], out["chains_check"]
    assert out["good_reads"] >= 0.9 * n_reads
    # every allele was reached by the given alignments: what recovery still carries over are the decoy alignments of 5 % of the reads
    assert out["alignments_transferred"] <= 0.1 * n_reads * out["alleles"]
    assert out["kept_after_filter"] >= 20 and out["annealing_chains"] == 400
    for k in ("score_reads_kernel", "prefilter_tile_kernel", "greedy_loop_kernel", "anneal_loop_kernel"):
        if k != "prefilter_tile_kernel" or out["alleles"] >= 100:        # run_filter is skipped when the first stage takes every genotype (solve.rs:940-945)
            assert out["kernel_ms"][k] > 0.0
    assert out["roofline"]["kernel"] in out["kernel_ms"]


def test_whole_path_on_a_small_locus(gpu_ctx):
    out = LR.ont_whole_path(gpu_ctx, 3000, n_alleles=24, chunk=500, read_len=6000, checker=CPU.whole_path_chains_check)
    check(out, 3000)
    # the same reads with their alignments counted by the caller, resident: same call (recovery adds the decoys' transfers in the
    # records form, so the two batches are not the same batch: the CALL is compared, and each form with the oracle on its own products)
    cnt = LR.ont_whole_path(gpu_ctx, 3000, n_alleles=24, chunk=500, read_len=6000, checker=CPU.whole_path_chains_check, counted=True)
    assert cnt["all_calls_equal_truth"] and cnt["chains_equal_oracle"] and cnt["alignments_transferred"] == 0
    assert cnt["called_genotype"] == out["called_genotype"]


def test_whole_path_at_65536_reads_and_256_alleles(gpu_ctx):
    """The configuration's allele count and read length at 1 / 15 of its reads: 16.8 M given alignments (45 GB of records and CIGAR words
    through the streaming batch), 32 896 genotypes prefiltered, 5 000 greedy and 400 annealing chains over 65 536 reads; eight chains of
    each kind against the oracle on the full batch."""
    out = LR.ont_whole_path(gpu_ctx, 65536, n_alleles=256, chunk=4096, checker=CPU.whole_path_chains_check)
    check(out, 65536)
    gpu_ctx.trim()
