#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03r
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_hp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_hp -- python $R/scripts/perf_hover_post.py 256 5 > $OUT/${TAG}_perf_hover_post.txt 2>&1
python $R/scripts/prof_summarize.py /tmp/rp_hp $OUT/${TAG}_hover_post_rocprofv3_summary.txt > /dev/null
grep -v amdgpu $OUT/${TAG}_perf_hover_post.txt | tail -2
head -50 $OUT/${TAG}_hover_post_rocprofv3_summary.txt | cut -c1-140
