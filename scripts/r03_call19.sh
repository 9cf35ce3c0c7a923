#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03m
cd $R
timeout 900 python -m pytest tests/test_engine.py tests/test_semantic.py tests/test_thin_head_gpu.py tests/test_hovernetplus.py -m gpu -q 2>&1 | tail -8
timeout 300 python scripts/perf_trunk.py 1024 256 2>&1 | grep -v "^No local" | tee $OUT/${TAG}_perf_trunk.txt
TIA_CONV_NO_SPATIAL=1 timeout 300 python scripts/perf_trunk.py 1024 256 2>&1 | grep -v "^No local" | tee $OUT/${TAG}_perf_trunk_nospatial.txt
timeout 300 python scripts/perf_conv.py 1024 2>&1 | grep -v "^No local" | tee $OUT/${TAG}_perf_conv.txt
timeout 300 python scripts/perf_hovernet_layers.py hovernet 32 2>&1 | head -12 | tee $OUT/${TAG}_hovernet_layers.txt
timeout 300 python scripts/perf_hovernet_layers.py unet 8 2>&1 | head -12 | tee $OUT/${TAG}_unet_layers.txt
