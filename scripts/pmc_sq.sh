#!/bin/bash
# SQ counter pass over a workload: pmc_sq.sh TAG STEM command...  -> gpurun_out/TAG_STEM_pmc_SQ.txt (per-kernel means per dispatch)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; STEM=$2; shift 2
mkdir -p $R/gpurun_out
CTRS="${SQ_COUNTERS:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU}"
rm -rf /tmp/rp_sq
(cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/rp_sq -- "$@" > /tmp/rp_sq.out 2>&1)
tail -5 /tmp/rp_sq.out; python $R/scripts/prof_summarize.py /tmp/rp_sq $R/gpurun_out/${TAG}_${STEM}_pmc_SQ.txt > /dev/null
grep -E "SQ_|GRBM" $R/gpurun_out/${TAG}_${STEM}_pmc_SQ.txt | cut -c1-150
