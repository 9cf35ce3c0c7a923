#!/bin/bash
# Round-3 GPU call 2: new parity tests (stem, full-size configs[2]/[4], HoVerNet+ tile mode, save_dir contracts), the whole
# GPU suite, rocprofv3 kernel trace of the bench and the --pmc traffic passes of the trunk and stain kernels (256x256).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
echo "== new tests"; timeout 900 python -m pytest tests/test_stem_gpu.py tests/test_fullsize_parity.py tests/test_hovernetplus.py tests/test_semantic.py tests/test_tile_mode.py -m gpu -q 2>&1 | tail -40 | tee $OUT/${TAG}_pytest_new.log
echo "== pytest -m gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $OUT/${TAG}_pytest_gpu.log
cd /tmp
echo "== rocprof kernel trace of the bench"
rm -rf /tmp/rp_bench; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_bench -- \
    python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
python $R/scripts/prof_summarize.py /tmp/rp_bench $OUT/${TAG}_bench_rocprofv3_summary.txt > /dev/null; head -16 $OUT/${TAG}_bench_rocprofv3_summary.txt | cut -c1-160
echo "== pmc passes"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_$c; timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/rp_$c -- \
      python $R/scripts/perf_trunk.py 1024 256 2 > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/rp_$c $OUT/${TAG}_trunk_pmc_${c}.txt > /dev/null
  rm -rf /tmp/rs_$c; timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/rs_$c -- \
      python $R/scripts/perf_stain.py 4096 256 > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/rs_$c $OUT/${TAG}_stain_pmc_${c}.txt > /dev/null
done
cat $OUT/${TAG}_trunk_pmc_FETCH_SIZE.txt $OUT/${TAG}_trunk_pmc_WRITE_SIZE.txt | cut -c1-150
cd $R
timeout 300 python scripts/perf_stain.py 4096 256 2>&1 | grep -v amdgpu | tee $OUT/${TAG}_perf_stain.txt
