#!/bin/bash
# Round-2 GPU call 5: convolution kernel v2 (buffer loads, per-slot tap masks) -- parity, per-layer timing, bench line.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
TAG=${1:-r02g}
echo "== conv tests"; timeout 600 python -m pytest tests/test_engine.py -m gpu -q -x -k "mfma or conv" 2>&1 | tail -5 | tee $OUT/${TAG}_pytest_conv.log
echo "== perf_conv"; timeout 300 python scripts/perf_conv.py 2>&1 | grep -v amdgpu | tee $OUT/${TAG}_perf_conv.txt
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 3000 $OUT/${TAG}_bench.json
