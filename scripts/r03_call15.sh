#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03i
cd $R
echo "== default (128 VGPRs)"; timeout 300 python scripts/perf_stain.py 4096 256 2>&1 | grep "^stats" | tee $OUT/${TAG}_perf_stain_default.txt
echo "== WPE=2 (256 VGPRs)"; TIA_LIB_PATH=$R/tiatoolbox_amd/lib/libtiatoolbox_amd_wpe2.so timeout 300 python scripts/perf_stain.py 4096 256 2>&1 | grep "^stats" | tee $OUT/${TAG}_perf_stain_wpe2.txt
echo "== WPE=2 224"; TIA_LIB_PATH=$R/tiatoolbox_amd/lib/libtiatoolbox_amd_wpe2.so timeout 300 python scripts/perf_stain.py 4096 224 2>&1 | grep "^stats"
