#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
echo "== perf_conv (rule)"; timeout 300 python scripts/perf_conv.py 1024 2>&1 | grep -v amdgpu | sed "s/| miopen.*//" | grep "1x1\|total"
for v in "" "TIA_CONV_NO_1X1_RULE=1"; do
echo "== fwd hovernet [$v]"; env $v timeout 300 python scripts/perf_hovernet_fwd.py 32 2>&1 | grep forward
done
