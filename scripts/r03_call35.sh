#!/bin/bash
# counters of the tap-reuse convolution kernel (separate --pmc passes over one trunk forward at the headline shape) + refreshed bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03p
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/rc_$name; timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/rc_$name -- python $R/scripts/perf_trunk.py 1024 256 > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/rc_$name $OUT/${TAG}_spatial_pmc_$name.txt > /dev/null
  grep "conv3x3_spatial_kernel<128, 0, (anonymous namespace)::G16" $OUT/${TAG}_spatial_pmc_$name.txt | cut -c1-60
done
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/${TAG}_bench.json
timeout 600 python -m pytest tests/test_engine.py tests/test_hovernet_post.py -m gpu -q 2>&1 | tail -3
