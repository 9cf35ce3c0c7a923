#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03s
cd $R
timeout 900 python -m pytest tests/test_hovernet_post.py tests/test_hovernetplus.py tests/test_tissuemask.py tests/test_tile_mode.py tests/test_fullsize_parity.py -m gpu -q -x 2>&1 | tail -8
timeout 300 python scripts/perf_hover_post.py 256 5 2>&1 | grep -v "amdgpu" | tail -1 | tee $OUT/${TAG}_perf_hover_post.txt
TIA_NO_CCL_TILE=1 timeout 300 python scripts/perf_hover_post.py 256 5 2>&1 | grep -v "amdgpu" | tail -1 | tee -a $OUT/${TAG}_perf_hover_post.txt
