#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03u
cd $R
timeout 900 python -m pytest tests/test_engine.py tests/test_semantic.py tests/test_thin_head_gpu.py -m gpu -q 2>&1 | tail -4
timeout 300 python scripts/perf_hovernet_layers.py unet 8 2>&1 | grep -v "amdgpu\|No local" > $OUT/${TAG}_unet_layers.txt; head -1 $OUT/${TAG}_unet_layers.txt
timeout 300 python scripts/perf_hovernet_layers.py hovernet 32 2>&1 | grep -v "amdgpu\|No local" > $OUT/${TAG}_hovernet_layers.txt; head -1 $OUT/${TAG}_hovernet_layers.txt
timeout 300 python scripts/perf_trunk.py 1024 256 2>&1 | grep -v "amdgpu\|No local" | tee $OUT/${TAG}_perf_trunk.txt
