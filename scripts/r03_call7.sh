#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03e
cd $R
echo "== half conv tests"; timeout 900 python -m pytest tests/test_engine.py tests/test_stem_gpu.py -m gpu -q -x 2>&1 | tail -25 | tee $OUT/${TAG}_pytest_half.log
echo "== perf"; timeout 600 python scripts/perf_conv_h.py 1024 256 float16 2>&1 | grep -v amdgpu | tee $OUT/${TAG}_perf_conv_h.txt
timeout 300 python scripts/perf_conv_h.py 1024 256 bfloat16 2>&1 | grep -v amdgpu | tail -1 | tee -a $OUT/${TAG}_perf_conv_h.txt
echo "== BK32"; TIA_CONVH_BK32=1 timeout 600 python scripts/perf_conv_h.py 1024 256 float16 2>&1 | grep -v amdgpu | tee $OUT/${TAG}_perf_conv_h_bk32.txt
