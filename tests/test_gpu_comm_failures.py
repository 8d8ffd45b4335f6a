"""Error paths of the multi-GPU exchanges (lcty_comm.hip): whatever a rank does on its own between two collectives runs behind a status
agreement every rank joins, so a failure on one rank makes every rank return an error instead of leaving the others inside RCCL.
With the one GPU of this box: a communicator of one rank, every agreement of every call failed in turn through the knob "comm_fail_at"
(the call returns the error, the communicator stays usable). With two or more GPUs: two processes, rank 1 fails at each agreement,
rank 0 must come back with "another rank ... failed" (RCCL refuses two ranks on one device, so that part needs two devices)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from locityper_amd import _lib, api, cdefs, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_agreement_of_every_exchange_can_fail_with_one_rank(gpu_ctx):
    L = synth.SynthLocus(12, 3000, seed=31, base_len=12000)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(gpu_ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    aa = api.AllAlignments.load(loc, L.reads(0, 3000))
    aa.prefilter_async()
    scores = aa.prefilter_scores()
    gts = api.generate_genotypes(12, 2)[:9]
    sv = api.default_solver(cdefs.SOLVER_GREEDY)
    seeds = api.chain_seeds(5, len(gts))
    want = api.solve_stage(aa, gts, sv, 1, seeds)[2]
    comm = api.Comm(gpu_ctx, 1, 0, api.comm_unique_id())
    assert comm.rccl_ranks() == (1, 0)
    gpu_ctx.set_knob("gather_chunk_mb", 1)                       # the rows travel in several chunks: an agreement per chunk
    calls = {"allreduce": lambda: comm.prefilter_allreduce(aa),
             "chains": lambda: comm.solve_stage(aa, gts, sv, 1, seeds)[2],
             "reads": lambda: comm.solve_stage_read_sharded(aa, gts, sv, 1, seeds)[2]}
    n_agreements = {}
    try:
        for name, call in calls.items():
            k = 1
            while True:
                gpu_ctx.set_knob("comm_fail_at", k)
                try:
                    got = call()
                except _lib.LocityperError as e:
                    assert e.code == cdefs.ERR_RUNTIME and "injected failure" in str(e)
                    k += 1
                    assert k < 200
                    continue
                break                                            # k is beyond the call's last agreement: it ran through
            n_agreements[name] = k - 1
            gpu_ctx.set_knob("comm_fail_at", -1)
            got = call()                                         # and the communicator is usable after every failure
            if name != "allreduce":
                assert np.array_equal(got, want)
    finally:
        gpu_ctx.set_knob("comm_fail_at", -1)
        gpu_ctx.set_knob("gather_chunk_mb", -1)
    assert np.array_equal(aa.prefilter_scores(), scores)
    assert n_agreements["allreduce"] == 1 and n_agreements["chains"] == 2 and n_agreements["reads"] >= 5, n_agreements
    comm.close()


@pytest.mark.parametrize("call,n_points", [("allreduce", 1), ("chains", 2), ("reads", 6)])
def test_a_failing_rank_releases_the_other_rank(tmp_path, call, n_points):
    if api.device_count() < 2:
        pytest.skip("RCCL refuses two ranks on one device: this part needs two GPUs")
    for fail_at in range(0, n_points + 1):
        id_file = str(tmp_path / f"id_{call}_{fail_at}")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "comm_fail_worker.py"), str(r), "2", id_file, call, "1", str(fail_at)],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
        outs = []
        for pr in procs:
            try:
                out, err = pr.communicate(timeout=600)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                pytest.fail(f"{call}: a rank stayed inside the exchange after rank 1 failed at agreement {fail_at}")
            assert pr.returncode == 0, err[-2000:]
            outs.append(out)
        if fail_at == 0:
            assert "rank 0: ok" in outs[0] and "rank 1: ok" in outs[1]
        else:
            assert "injected failure" in outs[1] and "another rank" in outs[0], outs


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("call", ["allreduce", "chains", "reads"])
def test_the_exchanges_across_real_ranks(tmp_path, call, world):
    """The RCCL exchanges with more than one rank, nothing failing: the two-rank (four-, eight-rank) forms of
    test_gpu_parity.py::test_read_sharded_prefilter_allreduce, ::test_chain_sharded_stage and ::test_read_sharded_stage. One FRESH
    process per rank (tests/comm_fail_worker.py; no process that has touched the GPU starts another program); every rank checks
    inside the worker that what the exchange gave it equals the single-device call on the whole batch — the all-reduced run_filter
    scores to 1e-11, the chain likelihoods of both sharded stages bit for bit. Skipped below `world` devices: RCCL refuses two ranks
    on one device, and every box of rounds 1-6 had one GPU — until a multi-GPU box runs this, nothing is claimed for RCCL beyond one rank."""
    if api.device_count() < world:
        pytest.skip(f"needs {world} GPUs (RCCL refuses several ranks on one device)")
    id_file = str(tmp_path / f"id_{call}_{world}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "comm_fail_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), id_file, call, "0", "0"], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    for r, pr in enumerate(procs):
        try:
            out, err = pr.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail(f"{call} over {world} ranks: rank {r} did not come back")
        assert pr.returncode == 0, err[-2000:]
        assert f"rank {r}: ok" in out, (out, err[-1500:])
