"""world_size-2 gloo test of the multi-GPU plumbing (runs on CPU): locus round-robin, max-over-ranks
timing, and the read-sharded run_filter formulation (partial scores of read shards SUM-all-reduced
== scores of the whole batch), with the oracle standing in for the per-shard computation."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np
from locityper_amd import dist, synth
from tests import oracle_ffi as O
rank, local_rank, world = dist.init("gloo")
assert world == 2
loci = dist.loci_for_rank(5, rank, world)
t = dist.max_over_ranks(1.0 + rank)
L = synth.SynthLocus(6, 400, seed=5, base_len=6000)
p = O.resolve_params(O.default_params(), L.bg)
ol = O.OracleLocus(L.seqs, L.seq_off, L.counts, L.cnt_off, L.k, L.bg, p)
ch = L.reads(0, 200)
gts = O.generate_genotypes(6, 2)
lo, hi = dist.read_shard(200, rank, world)
part = O.run_filter(ol.load(ch.slice(lo, hi)).best_aln_matrix(), gts)
tot = dist.allreduce_sum_f64(part)
full = O.run_filter(ol.load(ch).best_aln_matrix(), gts)
# level 3: the chains of a stage dealt to the ranks in contiguous genotype blocks, likelihoods all-gathered
oa = ol.load(ch)
sv = O.default_solver(1)
sv.anneal_steps, sv.plato_size = 300, 200
gsub = gts[:7]                                           # 7 genotypes over 2 ranks: blocks of 4 and 3
seeds = np.arange(3 * len(gsub), dtype=np.uint64) * 7919 + 11
clo, chi, per = dist.chain_block(len(gsub), rank, world)
local = O.solve_stage(ol, oa, gsub[clo:chi], sv, 3, seeds[3 * clo:3 * chi])[2]
liks = dist.allgather_chain_liks(local, len(gsub), 3)
whole = O.solve_stage(ol, oa, gsub, sv, 3, seeds)
# level 2 through the solver: every rank loaded only its shard of the reads; the per-pair products are all-gathered in rank order
# (the host-array form of lcty_solve_stage_read_sharded's exchange), then the chains are dealt as above
mine = ol.load(ch.slice(lo, hi))
st, w, unm, off, pa = dist.allgather_read_shards(mine.status, mine.weight, mine.unmapped_prob, mine.pa_off, mine.pair_alns)
joined = O.alns_from_arrays(6, st, w, unm, off, pa)
local2 = O.solve_stage(ol, joined, gsub[clo:chi], sv, 3, seeds[3 * clo:3 * chi])[2]
liks2 = dist.allgather_chain_liks(local2, len(gsub), 3)
dist.barrier()
res = dist.gather_objects({"rank": rank, "loci": loci, "shard": [lo, hi], "block": [clo, chi, per],
                            "read_sharded_equal": bool(np.array_equal(liks2, whole[2])), "joined_pairs": int(len(st)),
                            "chains_equal": bool(np.array_equal(liks, whole[2])),
                            "mean_equal": bool(np.array_equal(liks.mean(axis=1), whole[2].mean(axis=1)))})
if rank == 0:
    print(json.dumps({"t": t, "res": res, "maxrel": float(np.abs(tot - full).max() / np.abs(full).max()),
                      "argmax": [int(np.argmax(tot)), int(np.argmax(full))]}))
dist.finalize()
"""


def test_two_rank_gloo(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    import json
    out = json.loads(outs[0][0].strip().splitlines()[-1])
    assert out["t"] == 2.0
    assert sorted(out["res"][0]["loci"] + out["res"][1]["loci"]) == [0, 1, 2, 3, 4]
    assert out["res"][0]["shard"] == [0, 100] and out["res"][1]["shard"] == [100, 200]
    assert out["maxrel"] < 1e-13 and out["argmax"][0] == out["argmax"][1]
    # chains of a stage in blocks of ceil(7 / 2): the gathered likelihoods are the single-process ones, bit for bit
    assert out["res"][0]["block"] == [0, 4, 4] and out["res"][1]["block"] == [4, 7, 4]
    assert all(r["chains_equal"] and r["mean_equal"] for r in out["res"])
    # reads sharded, per-pair products exchanged: the same chains again
    assert all(r["read_sharded_equal"] and r["joined_pairs"] == 200 for r in out["res"])
