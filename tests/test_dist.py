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
dist.barrier()
res = dist.gather_objects({"rank": rank, "loci": loci, "shard": [lo, hi]})
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
