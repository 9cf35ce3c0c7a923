#!/bin/bash
# Round-3 GPU call 3: the register-resident stain statistics kernel: parity suite, timing, PMC traffic.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
echo "== stain tests"; timeout 900 python -m pytest tests/test_stain_gpu.py tests/test_fullsize_parity.py -m gpu -q -x 2>&1 | tail -30 | tee $OUT/${TAG}_pytest_stain.log
echo "== perf"; for hw in 224 256; do timeout 300 python scripts/perf_stain.py 4096 $hw 2>&1 | grep -v amdgpu | head -4; done | tee $OUT/${TAG}_perf_stain.txt
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rs_$c; timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/rs_$c -- \
      python $R/scripts/perf_stain.py 4096 224 > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/rs_$c $OUT/${TAG}_stain224_pmc_${c}.txt > /dev/null
  grep stain_stats $OUT/${TAG}_stain224_pmc_${c}.txt | cut -c1-130
done
cd $R
echo "== engine + others"; timeout 900 python -m pytest tests/test_engine.py tests/test_reinhard.py tests/test_tissuemask.py -m gpu -q 2>&1 | tail -5
