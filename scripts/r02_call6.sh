#!/bin/bash
# Round-2 GPU call 6: convolution kernel variants (TIA_CONV_VARIANT bit 0 double-buffered LDS, bit 1 s_setprio, bit 2 256x64 tiles)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for v in ${VARIANTS:-0 1 2 3 4 8}; do
  echo "== variant $v"
  TIA_CONV_VARIANT=$v timeout 300 python -m pytest tests/test_engine.py -m gpu -q -x -k "hip_mfma_conv" 2>&1 | tail -1
  TIA_CONV_VARIANT=$v timeout 300 python scripts/perf_conv.py 2>&1 | grep -v amdgpu | sed 's/| miopen.*//' | tee $OUT/r02h_perf_conv_v$v.txt
done
