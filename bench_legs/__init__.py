"""The measurement legs of bench.py (harness code: ctypes calls into liblocityper_hip.so, timing, the JSON line; nothing of the
path is computed here). bench.py itself keeps the argument list, the set-up of the workload, the timed region and the line.

    common      progress lines, host description, the constants of the rooflines
    launch      `bench.py --gpus N` without a launcher: one fresh child process per rank
    rooflines   per-kernel rooflines of a step, the committed counter passes beside them
    cpu         the CPU baseline (the oracle, built -march=native on the box it runs on) — the only module that touches tests.oracle_ffi
    loci_queue  the queue of loci that are not resident (uploads inside the steps)
    short_reads recruitment, candidate generation, alignment recovery, the 4 096-allele shard
    long_reads  BASELINE.json configs[2]: recovery, from bases, streamed, and the whole path on given alignments
"""
