#!/bin/bash
# MFMA-busy evidence for the hand-written convolution: separate rocprofv3 --pmc passes over scripts/perf_conv.py
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${1:-r02f}
cd /tmp && export TMPDIR=/tmp
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE; do
  rm -rf /tmp/rp_$c; timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/rp_$c -- \
      python $R/scripts/perf_conv.py 1024 > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/rp_$c $OUT/${TAG}_conv_pmc_${c}.txt > /dev/null 2>&1; grep -h conv_mfma $OUT/${TAG}_conv_pmc_${c}.txt | cut -c1-110
done
rm -rf /tmp/rp_kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_kt -- python $R/scripts/perf_conv.py 1024 > /dev/null 2>&1
python $R/scripts/prof_summarize.py /tmp/rp_kt $OUT/${TAG}_perf_conv_rocprofv3_summary.txt > /dev/null 2>&1; head -8 $OUT/${TAG}_perf_conv_rocprofv3_summary.txt | cut -c1-140
