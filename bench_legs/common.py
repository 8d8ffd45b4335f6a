"""Shared by the legs: progress lines on stderr, the host's cores, the constants the rooflines are priced against."""
import os
import sys
import time

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
VALU_F64_TMAXADD = 39.3        # 1 024 SIMDs x 16 lanes x 2.4 GHz: f64 max / add per second, in T
# SURVEY.md §8(d) / BASELINE.md algorithmic bytes per read pair scored (config 2, f64 matrix):
# 75 B packed bases + 2*A*16 B alignment table + 252*8 B k-mer probe slots + A*8 B matrix row
ALG_BYTES_FIXED = 75 + 2016

_T_START = time.time()


def survey_bytes_per_pair(n_alleles):
    return ALG_BYTES_FIXED + 2 * n_alleles * 16 + n_alleles * 8


def progress(what):
    """where the wall time of a run goes (stderr; the JSON line is the only thing on stdout)"""
    print(f"[bench {time.time() - _T_START:7.1f} s] {what}", file=sys.stderr, flush=True)


def physical_cores():
    """Physical cores of the host (unique (physical id, core id) pairs of /proc/cpuinfo); falls back to os.cpu_count()."""
    try:
        cores, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        n = len(cores) or os.cpu_count()
    except OSError:
        n = os.cpu_count()
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    return max(1, int(n))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def genotype_of(gts, ix):
    return tuple(int(x) for x in gts[int(ix)])
