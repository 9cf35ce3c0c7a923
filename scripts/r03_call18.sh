#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03l
cd $R
timeout 600 python -m pytest tests/test_engine.py -m gpu -q -k "half or conv" 2>&1 | tail -8
timeout 300 python scripts/perf_conv_h.py 1024 256 float16 2>&1 | grep -v "^No local" | tee $OUT/${TAG}_perf_conv_h_spatial.txt
TIA_CONVH_NO_SPATIAL=1 timeout 300 python scripts/perf_conv_h.py 1024 256 float16 2>&1 | grep "layer1 3x3\|layer2 3x3 \|layer3 3x3 \|trunk" | tee $OUT/${TAG}_perf_conv_h_nospatial.txt
timeout 300 python scripts/perf_conv_h.py 1024 224 float16 2>&1 | grep "layer1 3x3\|layer2 3x3 \|layer3 3x3 \|trunk" | tee $OUT/${TAG}_perf_conv_h_spatial224.txt
