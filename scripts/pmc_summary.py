"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, MI355X_MICROARCH.md §rocprofv3 PMC slots) into
profiles/<tag>_pmc_traffic.json: HBM bytes per launch of every lcty kernel.

  FETCH_SIZE, WRITE_SIZE are kilobytes (rocprofv3 -L). gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE tallies
  128-byte read requests at 64 bytes, i.e. reports half of a wide streaming read -> doubled here ("fetch_bytes_corrected");
  WRITE_SIZE is exact. Narrow gathers are uncalibrated, so both the raw and the corrected figure are kept.

usage: python scripts/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [workload note [read_pairs alleles]]
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    """(sum, launches) over the FULL-SIZE launches of every kernel: a dispatch counts when its grid is at least half the largest grid
    the kernel was launched with (the solver also launches its kernels on single chains, which would drag a plain mean down)."""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    grid = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        per[name][r["Dispatch_Id"]] += float(r["Counter_Value"])
        grid[name][r["Dispatch_Id"]] = float(r.get("Grid_Size") or 0)
    out = {}
    for k, d in per.items():
        biggest = max(grid[k].values())
        full = [i for i in d if grid[k][i] >= 0.5 * biggest]
        out[k] = (sum(d[i] for i in full), len(full))
    return out


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
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"note": sys.argv[4] if len(sys.argv) > 4 else "", "unit": "bytes per launch", "kernels": {}}
    if len(sys.argv) > 6:
        out["read_pairs"], out["alleles"] = int(sys.argv[5]), int(sys.argv[6])      # bench.py matches its workload against these
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("lcty::"):
            continue
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        fb = 1024.0 * f / max(nf, 1)
        wb = 1024.0 * w / max(nw, 1)
        out["kernels"][k] = {"launches": max(nf, nw), "fetch_bytes_raw": fb, "fetch_bytes_corrected": 2.0 * fb, "write_bytes": wb,
                             "hbm_bytes": 2.0 * fb + wb}
    stamp(out, sys.argv[1])
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:45s} x{v['launches']:<3d} fetch {v['fetch_bytes_corrected'] / 1e9:9.3f} GB (raw {v['fetch_bytes_raw'] / 1e9:8.3f})  "
              f"write {v['write_bytes'] / 1e9:8.3f} GB")


main()
