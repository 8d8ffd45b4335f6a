"""One step of the queue of loci out of a rocprofv3 kernel trace (scripts/gpu_round.sh <tag> p -> gpurun_out/<tag>/prof/v_kernel_trace.csv):
which kernel ran when on which stream, relative to the launch of the LAST full greedy loop but one.
   python3 scripts/queue_timeline.py <kernel_trace.csv> [min_ms]"""
import csv, sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Stream_Id"]), r["Kernel_Name"].replace("void ", "").replace("lcty::", "").split("(")[0],
           int(r["Grid_Size_X"]), int(r["LDS_Block_Size"])) for r in rows]
    greedy = [e for e in ev if e[3].startswith("greedy_loop_kernel")]
    big = max(e[4] for e in greedy)
    full = [e for e in greedy if e[4] == big]
    if len(full) < 3:
        sys.exit("fewer than three full-size greedy launches in the trace")
    t0, prev, nxt = full[-2][0], full[-3][0], full[-1][0]
    print(f"step (greedy launch to greedy launch): {(nxt - t0) / 1e6:.1f} ms; the step before: {(t0 - prev) / 1e6:.1f} ms")
    print("    start ..      end     stream  duration  LDS/WG  kernel")
    for s, e, st, name, grid, lds in sorted(ev):
        if e < prev + (t0 - prev) * 0.75 or s > nxt + 10_000_000 or (e - s) / 1e6 < min_ms:
            continue
        print(f"  {(s - t0) / 1e6:8.1f} .. {(e - t0) / 1e6:8.1f} ms  {st:3d}  {(e - s) / 1e6:8.1f} ms  {lds:6d}  {name}  (grid {grid})")


if __name__ == "__main__":
    main()
