#!/bin/bash
# Round-2 GPU call 4: window-selection stain_stats -- parity (incl. the bitwise audit vs the histogram path), timing,
# per-phase cycles, and LDS / VALU counters of the kernel (own --pmc pass, kernel-trace only).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
echo "== stain tests"; timeout 900 python -m pytest tests/test_stain_gpu.py tests/test_engine.py -m gpu -q -x 2>&1 | tail -25 | tee $OUT/r02d_pytest_stain.log
echo "== perf_stain 224"; timeout 300 python scripts/perf_stain.py 4096 2>&1 | grep -v amdgpu | tee $OUT/r02d_perf_stain.txt
echo "== perf_stain 256"; timeout 300 python scripts/perf_stain.py 4096 256 2>&1 | grep -v amdgpu | head -3 | tee -a $OUT/r02d_perf_stain.txt
cd /tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/rp_$tag; timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/rp_$tag -- python $R/scripts/perf_stain.py 4096 > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/rp_$tag $OUT/r02d_stain_pmc_${tag}.txt > /dev/null; grep -h "stain_stats" $OUT/r02d_stain_pmc_${tag}.txt | cut -c1-110
done
