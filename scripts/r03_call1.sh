#!/bin/bash
# Round-3 GPU call 1: the new stem kernel (tests + timing), the whole GPU suite, the default bench at 256x256.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== stem tests"; timeout 600 python -m pytest tests/test_stem_gpu.py -m gpu -q -x 2>&1 | tail -25 | tee $OUT/r03a_pytest_stem.log
echo "== perf_trunk"; timeout 300 python scripts/perf_trunk.py 1024 256 2>&1 | grep -v amdgpu | tee $OUT/r03a_perf_trunk.txt
timeout 300 python scripts/perf_trunk.py 1024 224 2>&1 | grep -v amdgpu | tee -a $OUT/r03a_perf_trunk.txt
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $OUT/r03a_pytest_gpu.log
echo "== bench"; timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/r03a_bench.json 2> $OUT/r03a_bench.err; echo "bench rc=$?"; tail -c 5000 $OUT/r03a_bench.json; tail -5 $OUT/r03a_bench.err
