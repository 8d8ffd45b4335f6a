"""bench.py as the driver starts it: `python3 bench.py --gpus N` with no launcher must put N ranks on the job by itself (children
started before the parent touches the GPU) and say so in its line; with a launcher's environment it runs as one of the ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--pairs", "20000", "--alleles", "16", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--ont-sample", "0",
         "--many-alleles-sample", "0", "--recruit-sample", "0", "--map-sample", "0", "--recovery-sample", "0", "--ont-stream-sample", "0", "--ont-whole-path-sample", "0", "--exact-sample", "0"]


def run_bench(extra, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, capture_output=True, text=True, env=env, timeout=1500)


def test_self_launch_without_a_gpu_fails_loudly():
    """CPU tier: the parent starts its two children, they find no device, the parent exits non-zero and prints no line."""
    from locityper_amd import api
    if api.device_count() > 0:
        pytest.skip("a GPU is visible: the launch is covered by the gpu-marked test below")
    r = run_bench(["--gpus", "2"])
    assert r.returncode != 0 and r.stdout.strip() == "" and "child ranks exited" in r.stderr


@pytest.mark.gpu
def test_self_launch_reports_the_ranks_it_started():
    from locityper_amd import api
    ndev = api.device_count()
    over = [] if ndev >= 2 else ["--oversubscribe"]
    r = run_bench(["--gpus", "2"] + over)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["launch"].startswith("self-spawned") and out["devices_used"] == min(2, ndev)
    assert out["solver"]["all_calls_equal_truth"] is True
    # more ranks than devices without the explicit flag: refused, non-zero, no line
    if ndev < 2:
        r2 = run_bench(["--gpus", "2"])
        assert r2.returncode != 0 and r2.stdout.strip() == ""
    # one rank: the plain single-process form
    r1 = run_bench(["--gpus", "1"])
    assert r1.returncode == 0, r1.stderr[-3000:]
    o1 = json.loads(r1.stdout.strip().splitlines()[-1])
    assert o1["n_gpus"] == 1 and o1["launch"] == "single process" and o1["rccl_ranks"] is None


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["--shard-reads", "--shard-chains"])
def test_one_locus_sharded_over_the_ranks_through_the_whole_path(mode):
    """BASELINE configs[4]'s form (`--shard-reads`: the reads of ONE locus over the ranks — run_filter scores all-reduced, location-table
    rows all-gathered per solver stage, chains dealt to the ranks) and the chain-sharded form, solver on, through the launch path of the
    driver. RCCL refuses two ranks on one device, so a one-GPU box runs the communicator with one rank; two devices run two."""
    from locityper_amd import api
    n = 2 if api.device_count() >= 2 else 1
    r = run_bench(["--gpus", str(n), mode])
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == n and out["rccl_ranks"] == n and out["scaling"] == "strong"
    assert out["solver"] is not None and tuple(out["called_genotype"]) == tuple(out["true_genotype"])
    if n > 1:
        assert out["ms_per_step_ranks"]["min"] <= out["ms_per_step_ranks"]["max"] == pytest.approx(out["ms_per_step"])
