# the long route of candidate generation on the GPU box: its tests, then the timing probe (scripts/map_long_probe.py)
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${1:-long}
mkdir -p $OUT
timeout 1500 python3 -m pytest tests/test_gpu_map_long.py tests/test_gpu_map.py -q > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -25 $OUT/pytest.log
timeout 900 python3 scripts/map_long_probe.py --reads 2048 --alleles 16 > $OUT/probe_8.json 2> $OUT/probe_8.err; cat $OUT/probe_8.json; tail -3 $OUT/probe_8.err
timeout 900 python3 scripts/map_long_probe.py --reads 1024 --alleles 64 --base-len 30000 > $OUT/probe_64.json 2> $OUT/probe_64.err; cat $OUT/probe_64.json; tail -3 $OUT/probe_64.err
