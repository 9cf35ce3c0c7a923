#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03f
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/rc_$name; timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/rc_$name -- python $R/scripts/perf_conv_h_one.py 1024 128 128 32 3 1 10 > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/rc_$name $OUT/${TAG}_convh_pmc_$name.txt > /dev/null
  grep conv_mfma_h $OUT/${TAG}_convh_pmc_$name.txt | cut -c1-70
done
cd $R
timeout 900 python -m pytest tests/test_engine.py -m gpu -q 2>&1 | tail -4
