#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03d
cd $R
for hw in 224 256; do TIA_LIB_PATH=$R/tiatoolbox_amd/lib/libtiatoolbox_amd_timing.so timeout 300 python scripts/perf_stain.py 4096 $hw 2>&1 | grep -v amdgpu | head -4; done | tee $OUT/${TAG}_perf_stain_timing.txt
for hw in 224 256; do timeout 300 python scripts/perf_stain.py 4096 $hw 2>&1 | grep -v amdgpu | head -2; done | tee $OUT/${TAG}_perf_stain.txt
timeout 600 python -m pytest tests/test_stain_gpu.py -m gpu -q -x 2>&1 | tail -3
