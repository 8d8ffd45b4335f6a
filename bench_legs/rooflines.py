"""Per-kernel rooflines of a step of the main measurement, with the committed counter passes beside them (DESIGN.md §4 for the bytes)."""
import glob
import json
import os
import re

from locityper_amd import api
from .common import HBM_PEAK_GBS, VALU_F64_TMAXADD, survey_bytes_per_pair

KERNEL_IDS = (("score_reads_kernel", "K_SCORE"), ("prefilter_tile_kernel", "K_PREFILTER"), ("solve_init_kernel", "K_SOLVE_INIT"),
              ("solve_init_kernel_annealing_stage", "K_SOLVE_INIT_ANNEAL"), ("greedy_loop_kernel", "K_SOLVE"),
              ("anneal_loop_kernel", "K_ANNEAL"), ("build_loc_table_kernel", "K_SOLVE_TABLE"))


def read_timers(ctx):
    """{kernel: (launches, total ms)} of the context's event timers since the last reset."""
    return {name: ctx.timing(getattr(api, kid)) for name, kid in KERNEL_IDS}


def newest(root, pattern):
    files = sorted(glob.glob(os.path.join(root, "profiles", pattern)))
    return files[-1] if files else None


def gather_ceiling(root):
    """Random 32-byte gathers per second out of a 148 GB footprint, from the newest committed run of scripts/gather_probe.hip."""
    path = newest(root, "r*_gather_probe.txt")
    best = None
    if path:
        for line in open(path):
            m = re.search(r"window\s+0 MB.*?([0-9.]+) G gathers/s", line)
            if m:
                best = max(best or 0.0, float(m.group(1)) * 1e9)
    return best, (os.path.relpath(path, root) if path else None)


def kernel_rooflines(root, timers, kern_steps, n_pairs, n_alleles, n_genotypes, n_good, per_step, beside_the_chains):
    """One entry per kernel of a step. `bytes` are ALGORITHMIC bytes (SURVEY §8(d) per-unit figures x the units of a launch), not traffic:
    the counter passes give that (attach_traffic)."""
    def ms(name):
        return timers[name][1]

    def launch_ms(name):
        n, t = timers[name]
        return t / max(n, 1)

    alg_bytes = survey_bytes_per_pair(n_alleles) * n_pairs
    chains_step = per_step["greedy_chains"] + per_step["anneal_chains"]
    chain_bytes = 34.0 * n_good + 207e3
    roofs = {
        "score_reads_kernel": {"bound": "hbm", "bytes": alg_bytes, "what": "SURVEY 8(d): 75 + 2*A*16 + 2016 + A*8 B per read pair"},
        "prefilter_tile_kernel": {"bound": "valu_f64", "ops": 2.0 * n_genotypes * n_pairs, "what": "2 * G * R max-add"},
        # the two initialisations of a step apart: the greedy stage's chains (~5 000, main stream) and the annealing stage's (400, side stream)
        "solve_init_kernel": {
            "bound": "hbm", "bytes": chain_bytes * per_step["greedy_chains"] if chains_step else 0.0,
            "chains_per_launch": per_step["greedy_chains"],
            "what": "solve_init_tile_kernel on the chains of the greedy stage; SURVEY 8(d): the reads CSR once per genotype x attempt, 34 B * R "
                    "+ 207 KB LUT per chain — a model figure: the kernel reads a table row once per GROUP of chains and writes 32 B per "
                    "non-trivial read; its real traffic is in `traffic`"},
        "solve_init_kernel_annealing_stage": {
            "bound": "hbm", "bytes": chain_bytes * per_step["anneal_chains"] if chains_step else 0.0,
            "chains_per_launch": per_step["anneal_chains"],
            "what": "the same kernel on the chains of the annealing stage (side stream, beside the next locus); same model figure"},
        "greedy_loop_kernel": {"bound": "hbm", "bytes": 32.0 * 10 * per_step["greedy_iterations"],
                               "what": "one 32 B record per candidate read, 10 candidates per iteration"},
        "anneal_loop_kernel": {"bound": "hbm", "bytes": 32.0 * per_step["anneal_moves"],
                               "what": "one 32 B record per evaluated move (latency-bound serial chains)"},
    }
    for name, r in roofs.items():
        r["ms_per_step"] = ms(name) / kern_steps
        r["launch_ms"] = launch_ms(name)
        r["bytes_are"] = "algorithmic (SURVEY 8d), not traffic"
    if beside_the_chains:
        # lcty_solve_queue issues everything before the chains of a locus on a third stream, beside the greedy chains of the locus before
        # (one wavefront per SIMD, 125 of 160 KB of LDS): launch_ms of these two is the kernel in THAT place — off the critical path of
        # a step — and launch_ms_alone / frac_alone the kernel with the device to itself
        for name in ("score_reads_kernel", "prefilter_tile_kernel"):
            roofs[name]["in_the_queue"] = "fore stream, beside the greedy chains of the locus before; not on the critical path of a step"
    # the loop kernels are random 32-byte gathers out of the chains' 148 GB of records: what the device does of THOSE at best is the
    # ceiling their record gathers are held against; SURVEY 8(d) itself calls K14 latency-bound
    ceiling, ceiling_src = gather_ceiling(root)
    for name, gathers in (("greedy_loop_kernel", 10.0 * per_step["greedy_iterations"]), ("anneal_loop_kernel", per_step["anneal_moves"])):
        r = roofs[name]
        if r["launch_ms"] > 0 and ceiling:
            r["record_gathers_per_s"] = gathers / (r["launch_ms"] * 1e-3)
            r["gather_ceiling_frac"] = r["record_gathers_per_s"] / ceiling
            r["gather_ceiling"] = {"gathers_per_s": ceiling, "source": f"{ceiling_src} (scripts/gather_probe.hip: 148 GB footprint)"}
    for r in roofs.values():
        if r["bound"] == "hbm":
            r["achieved"] = r["bytes"] / max(r["launch_ms"], 1e-9) / 1e6
            r["peak"], r["unit"] = HBM_PEAK_GBS, "GB/s"
        else:
            r["achieved"] = r["ops"] / max(r["launch_ms"], 1e-9) / 1e9
            r["peak"], r["unit"] = VALU_F64_TMAXADD, "Tmaxadd/s"
        r["frac"] = r["achieved"] / r["peak"]
    return roofs


def attach_alone(roofs, alone_ms):
    """The same kernels of one more locus solved call by call, nothing else on the device."""
    for name, r in roofs.items():
        if alone_ms.get(name, 0) > 0:
            r["launch_ms_alone"] = alone_ms[name]
            if "bytes" in r:
                r["frac_alone"] = r["bytes"] / (alone_ms[name] * 1e-3) / 1e9 / HBM_PEAK_GBS


def attach_traffic(root, roofs, path, n_pairs, n_alleles, sha16):
    """HBM traffic from the committed PMC passes (counters cannot be read from inside this process). FETCH_SIZE on gfx950 counts a
    128-byte read request as 64 bytes (MI355X_MICROARCH.md): doubled for the kernels that STREAM wide coalesced reads; kernels that
    gather 8-32 bytes per lane are outside that calibration and keep the raw figure. Both are in the line. Returns what the line's
    `roofline` object says about the file, or {}."""
    streaming = {"score_reads_kernel", "solve_init_kernel", "prefilter_tile_kernel"}
    try:
        tr = json.load(open(path))
        if tr.get("read_pairs") != n_pairs or tr.get("alleles") != n_alleles:
            return {}
        for name, r in roofs.items():
            pat = name.replace("score_reads_kernel", "score_").replace("solve_init_kernel", "solve_init_tile_kernel")
            cands = [v for n, v in tr["kernels"].items() if pat in n]
            k = max(cands, key=lambda v: v.get("fetch_bytes_raw", 0.0) + v.get("write_bytes", 0.0)) if cands else None
            if not k:
                continue
            f = 2.0 if name in streaming else 1.0
            r["traffic_fetch_raw"] = k.get("fetch_bytes_raw")
            r["traffic_fetch_x2"] = 2.0 * k.get("fetch_bytes_raw", 0.0)
            r["traffic_write"] = k.get("write_bytes")
            r["traffic_rule"] = "2 x FETCH_SIZE + WRITE_SIZE (streaming reads)" if f == 2.0 else "FETCH_SIZE + WRITE_SIZE (narrow gathers: raw)"
            r["traffic"] = f * k.get("fetch_bytes_raw", 0.0) + k.get("write_bytes", 0.0)
            if r.get("launch_ms"):
                r["traffic_GBs"] = r["traffic"] / (r["launch_ms"] * 1e-3) / 1e9        # what the launch really moved per second
        return {"traffic_source": os.path.relpath(path, root), "traffic_taken_at_commit": tr.get("taken_at_commit"),
                "traffic_sources_sha16": tr.get("sources_sha16"), "traffic_is_current": tr.get("sources_sha16") == sha16}
    except (OSError, KeyError, ValueError, TypeError):
        return {}


def attach_sq(root, roofs, sha16):
    """What the wavefronts of each kernel were doing (committed SQ counter pass, scripts/pmc_sq_summary.py): the fraction of their cycles
    with an instruction in flight / waiting. issuing x wavefronts per SIMD near or above 1 = bound by the instruction stream."""
    path = newest(root, "r*_pmc_sq.json")
    try:
        doc = json.load(open(path))
        sq = doc["kernels"]
    except (OSError, KeyError, ValueError, TypeError):
        return
    per_simd = {"greedy_loop_kernel": 1.0, "solve_init_kernel": 2.0, "score_reads_kernel": 4.0, "anneal_loop_kernel": 0.8}
    for name, r in roofs.items():
        pat = name.replace("score_reads_kernel", "score_counted_lean").replace("solve_init_kernel", "solve_init_tile_kernel")
        cands = [v for n, v in sq.items() if pat in n]
        if not cands:
            continue
        k = max(cands, key=lambda v: v.get("SQ_WAVE_CYCLES", 0.0))
        r["sq"] = {"issuing_frac": round(k.get("active_inst_frac", 0.0), 3), "waiting_frac": round(k.get("wait_any_frac", 0.0), 3),
                   "wavefronts_per_simd": per_simd.get(name), "source": os.path.relpath(path, root),
                   "is_current": doc.get("sources_sha16") == sha16}
        # vector instructions issued per launch against what the SIMDs could issue in the launch's time (one per 4 cycles each); a kernel
        # that runs beside the greedy chains in the queue: against its time with the device to itself
        launches = max(k.get("launches", 0.0), 1.0)
        ms_for_issue = r.get("launch_ms_alone") if r.get("in_the_queue") and r.get("launch_ms_alone") else r.get("launch_ms")
        if k.get("SQ_INSTS_VALU") and ms_for_issue:
            r["sq"]["valu_issue_frac"] = (k["SQ_INSTS_VALU"] / launches) / (1024 * 2.4e9 / 4.0 * ms_for_issue * 1e-3)
