"""Turns a rocprofv3 --pmc SQ_* pass (scripts/gpu_round.sh <tag> q) into profiles/<tag>_pmc_sq.json: the counters summed over the launches
of every lcty kernel and two fractions of SQ_WAVE_CYCLES — wait_any_frac (SQ_WAIT_ANY) and active_inst_frac (SQ_ACTIVE_INST_ANY).

usage: python scripts/pmc_sq_summary.py <counter_collection.csv> <out.json> [note]"""
import collections
import csv
import json
import sys


def stamp(doc, counter_file):
    """Which sources the pass saw (sources.sha16 that scripts/gpu_round.sh left in the pass's directory tree) and the commit this summary is
    made at (the pass is run on a snapshot of the committed tree)."""
    import os, subprocess
    d = os.path.dirname(os.path.abspath(counter_file))
    for _ in range(6):
        f = os.path.join(d, "sources.sha16")
        if os.path.exists(f):
            doc["sources_sha16"] = open(f).read().strip()
            break
        d = os.path.dirname(d)
    try:
        doc["taken_at_commit"] = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
    except Exception:
        pass


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    # per kernel, over its FULL-SIZE launches only (grid at least half its largest: the solver also launches its kernels on a few chains)
    disp = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    grid = collections.defaultdict(dict)
    for r in csv.DictReader(open(src)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if "lcty" not in name:
            continue
        disp[name][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        grid[name][r["Dispatch_Id"]] = float(r.get("Grid_Size") or 0)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for name, d in disp.items():
        biggest = max(grid[name].values())
        full = [i for i in d if grid[name][i] >= 0.5 * biggest]
        for i in full:
            for c, v in d[i].items(): per[name][c] += v
        per[name]["launches"] = float(len(full))
    out = {}
    for k, c in per.items():
        d = dict(sorted(c.items()))
        wc = d.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            d["wait_any_frac"] = d.get("SQ_WAIT_ANY", 0.0) / wc
            d["active_inst_frac"] = d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        out[k] = d
    doc = {"note": note, "kernels": out}
    stamp(doc, src)
    json.dump(doc, open(dst, "w"), indent=1)
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0))[:8]:
        print(f"{k[:60]:60s} issuing {d.get('active_inst_frac', 0):.2f} waiting {d.get('wait_any_frac', 0):.2f} VALU {d.get('SQ_INSTS_VALU', 0):.3g} SALU {d.get('SQ_INSTS_SALU', 0):.3g}")


main()
