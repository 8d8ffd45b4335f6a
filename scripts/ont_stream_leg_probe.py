"""The configs[2] leg of bench.py alone (locityper_amd/legs.py): python3 scripts/ont_stream_leg_probe.py [reads] [chunk]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from locityper_amd import api, legs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = api.Context(0)
print(json.dumps(legs.ont_from_bases_stream(ctx, n, 256, chunk=chunk, progress=lambda s: print(s, file=sys.stderr, flush=True))[0]))
