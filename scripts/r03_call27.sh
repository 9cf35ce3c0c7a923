#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03t
cd $R
timeout 900 python -m pytest tests/test_hovernet_post.py tests/test_hovernetplus.py tests/test_tissuemask.py tests/test_tile_mode.py tests/test_fullsize_parity.py -m gpu -q -x 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_hp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_hp -- python $R/scripts/perf_hover_post.py 256 5 > $OUT/${TAG}_perf_hover_post.txt 2>&1
python $R/scripts/prof_summarize.py /tmp/rp_hp $OUT/${TAG}_hover_post_rocprofv3_summary.txt > /dev/null
grep "proc_np_hv" $OUT/${TAG}_perf_hover_post.txt
head -16 $OUT/${TAG}_hover_post_rocprofv3_summary.txt | cut -c1-130
cd $R
timeout 300 python scripts/perf_hover_post.py 256 5 2>&1 | grep "proc_np_hv" | tee -a $OUT/${TAG}_perf_hover_post.txt
