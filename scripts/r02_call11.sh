#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_hf; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_hf -- python $R/scripts/perf_hovernet_fwd.py 32 2>&1 | grep -v amdgpu | tail -2
python $R/scripts/prof_summarize.py /tmp/rp_hf $OUT/r02t_hovernet_fwd_rocprofv3_summary.txt > /dev/null; head -32 $OUT/r02t_hovernet_fwd_rocprofv3_summary.txt | cut -c1-160
