"""Turns a rocprofv3 --pmc SQ_* pass (scripts/gpu_round.sh <tag> q) into profiles/<tag>_pmc_sq.json: the counters summed over the launches
of every lcty kernel and two fractions of SQ_WAVE_CYCLES — wait_any_frac (SQ_WAIT_ANY) and active_inst_frac (SQ_ACTIVE_INST_ANY).

usage: python scripts/pmc_sq_summary.py <counter_collection.csv> <out.json> [note]"""
import collections
import csv
import json
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(src)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if "lcty" not in name:
            continue
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
    out = {}
    for k, c in per.items():
        d = dict(sorted(c.items()))
        wc = d.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            d["wait_any_frac"] = d.get("SQ_WAIT_ANY", 0.0) / wc
            d["active_inst_frac"] = d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        out[k] = d
    json.dump({"note": note, "kernels": out}, open(dst, "w"), indent=1)
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0))[:8]:
        print(f"{k[:60]:60s} issuing {d.get('active_inst_frac', 0):.2f} waiting {d.get('wait_any_frac', 0):.2f} VALU {d.get('SQ_INSTS_VALU', 0):.3g} SALU {d.get('SQ_INSTS_SALU', 0):.3g}")


main()
