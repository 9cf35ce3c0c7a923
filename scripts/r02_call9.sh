#!/bin/bash
# Round-2 GPU call 9: counters of the convolution kernel (separate --pmc passes, kernel trace only): MFMA busy, wait
# breakdown, LDS, HBM traffic.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${1:-r02l}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/rp_$i; timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/rp_$i -- python $R/scripts/perf_conv.py 1024 > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/rp_$i $OUT/${TAG}_conv_pmc_set$i.txt > /dev/null 2>&1
  echo "== set $i: $set"; grep -h "conv_mfma" $OUT/${TAG}_conv_pmc_set$i.txt | cut -c1-125
done
