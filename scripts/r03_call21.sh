#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03o
cd $R
timeout 900 python -m pytest tests/test_engine.py tests/test_semantic.py tests/test_thin_head_gpu.py -m gpu -q 2>&1 | tail -6
timeout 300 python scripts/perf_trunk.py 1024 256 2>&1 | grep -v "amdgpu\|No local" | tee $OUT/${TAG}_perf_trunk.txt
timeout 300 python scripts/perf_conv_h.py 1024 256 float16 2>&1 | grep "layer4 3x3 \|trunk" | tee $OUT/${TAG}_perf_conv_h.txt
