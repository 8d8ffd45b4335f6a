"""configs[2] from bases on a basis of the alleles (locityper_amd/legs.py): python3 scripts/ont_basis_probe.py [--reads N] [--alleles A] [--basis B] [--chunk C]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locityper_amd import api, legs

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=8192)
ap.add_argument("--alleles", type=int, default=256)
ap.add_argument("--basis", type=int, default=16)
ap.add_argument("--chunk", type=int, default=2048)
a = ap.parse_args()
ctx = api.Context(0)
out, _ = legs.ont_from_bases_on_a_basis(ctx, a.reads, a.alleles, a.basis, a.chunk, progress=lambda s: print(s, file=sys.stderr, flush=True))
print(json.dumps(out))
