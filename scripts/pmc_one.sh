#!/bin/bash
# One FETCH_SIZE / WRITE_SIZE pass pair over a workload: pmc_one.sh TAG STEM command...  -> gpurun_out/TAG_STEM_pmc_{FETCH,WRITE}_SIZE.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; STEM=$2; shift 2
mkdir -p $R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp1_$c
  (cd /tmp && TMPDIR=/tmp timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/rp1_$c -- "$@" > /dev/null 2>&1)
  python $R/scripts/prof_summarize.py /tmp/rp1_$c $R/gpurun_out/${TAG}_${STEM}_pmc_${c}.txt | grep "$c" | cut -c1-120
done
