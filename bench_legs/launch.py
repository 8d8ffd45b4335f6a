"""`python3 bench.py --gpus N` without a launcher's environment: this process becomes the launcher."""
import os
import socket
import subprocess
import sys
import threading
import time

from .common import physical_cores


def spawn_ranks(args, script):
    """Starts N fresh children of this very command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, BEFORE the
    process has loaded the HIP library or made any GPU call (a process that has touched the GPU never execs another program), relays
    rank 0's JSON line and exits non-zero when any child fails. The children are what `torchrun --nproc-per-node N` would have started."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                LCTY_BENCH_LAUNCH="self-spawned children of bench.py")
    # every rank's OpenMP teams (synthetic data, CSR validation), loader and validation threads get their share of the host's cores:
    # N ranks with the default "all cores" each would oversubscribe the host N-fold during set-up and inside the loader threads
    share = max(1, physical_cores() // max(args.gpus, 1))
    if "OMP_NUM_THREADS" not in os.environ:
        base["OMP_NUM_THREADS"] = str(share)
    base["LCTY_BENCH_HOST_THREADS"] = str(share)
    cmd = [sys.executable, os.path.abspath(script)] + sys.argv[1:]
    procs = []
    for r in range(args.gpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    # rank 0's line is read by a thread; the launcher polls ALL children: one that dies before the rendezvous would leave the others
    # waiting for it for ever — the rest is ended and the launcher exits non-zero as soon as any child fails
    got = {}
    reader = threading.Thread(target=lambda: got.setdefault("line", procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            time.sleep(2.0)                                        # the others may be on their way out with the same error
            for i, p in enumerate(procs):
                if codes[i] is None and p.poll() is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            break
        time.sleep(0.2)
    if any(codes):
        print(f"bench.py: child ranks exited with {codes}", file=sys.stderr)
        sys.exit(next(c for c in codes if c) or 1)
    reader.join(timeout=30)
    sys.stdout.write(got.get("line", ""))
    sys.stdout.flush()
    sys.exit(0)
