"""Worker of tests/test_gpu_comm_failures.py: one rank of an RCCL communicator (one process per GPU). Rank `fail_rank` fails the
`fail_at`-th status agreement of the call under test (knob "comm_fail_at"); every rank must come back from the call with an error
(exit code 0 here and a line "rank r: error <code>: <message>"), none may stay inside RCCL. fail_at = 0: nothing fails, the call must succeed.
usage: comm_fail_worker.py <rank> <world> <id-file> <call> <fail_rank> <fail_at>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, id_file, call, fail_rank, fail_at = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    import numpy as np
    from locityper_amd import _lib, api, cdefs, synth
    os.environ["NCCL_DEBUG_FILE"] = os.devnull
    ctx = api.Context(rank % api.device_count())
    if rank == 0:
        uid = api.comm_unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > 120:
                raise RuntimeError("no communicator id from rank 0")
            time.sleep(0.05)
        uid = open(id_file, "rb").read()
    comm = api.Comm(ctx, world, rank, uid)
    assert comm.rccl_ranks() == (world, rank)
    L = synth.SynthLocus(12, 3000, seed=31, base_len=12000)
    p = api.resolve_params(api.default_params(), L.bg)
    loc = api.Locus(ctx, L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
    ch = L.reads(0, 3000)
    per = (3000 + world - 1) // world
    whole = api.AllAlignments.load(loc, ch)
    shard = api.AllAlignments.load(loc, ch.slice(min(rank * per, 3000), min((rank + 1) * per, 3000)))
    gts = api.generate_genotypes(12, 2)[:9]
    sv = api.default_solver(cdefs.SOLVER_GREEDY)
    seeds = api.chain_seeds(5, len(gts))
    want = api.solve_stage(whole, gts, sv, 1, seeds)[2]
    if rank == fail_rank and fail_at > 0:
        ctx.set_knob("comm_fail_at", fail_at)
    try:
        if call == "allreduce":
            shard.prefilter_async()
            comm.prefilter_allreduce(shard)
            whole.prefilter_async()
            w = whole.prefilter_scores()
            assert np.abs(shard.prefilter_scores() - w).max() <= 1e-11 * np.abs(w).max()
        elif call == "chains":
            got = comm.solve_stage(whole, gts, sv, 1, seeds)[2]
            assert np.array_equal(got, want)
        elif call == "reads":
            ctx.set_knob("gather_chunk_mb", 1)
            got = comm.solve_stage_read_sharded(shard, gts, sv, 1, seeds)[2]
            assert np.array_equal(got, want)
        else:
            raise ValueError(call)
        print(f"rank {rank}: ok", flush=True)
    except _lib.LocityperError as e:
        print(f"rank {rank}: error {e.code}: {e}", flush=True)


if __name__ == "__main__":
    main()
