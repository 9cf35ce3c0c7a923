#!/bin/bash
# Round-end measurement pass on the GPU box (everything lands in gpurun_out/, copied to profiles/ afterwards):
# full GPU test suite, smoke, the default bench (+cpu_baseline), rocprofv3 kernel trace of the bench, separate --pmc
# passes (HBM traffic of the stain kernels; the convolution kernel's counters are scripts/r02_call9.sh), the other kernel
# families, the other bench configurations.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${1:-r02p}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/${TAG}_pytest_gpu.log; cat $OUT/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-600 $OUT/${TAG}_bench.json
cd /tmp
rm -rf /tmp/rp_bench; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_bench -- \
    python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
python $R/scripts/prof_summarize.py /tmp/rp_bench $OUT/${TAG}_bench_rocprofv3_summary.txt > /dev/null; head -14 $OUT/${TAG}_bench_rocprofv3_summary.txt | cut -c1-150
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_$c; timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/rp_$c -- \
      python $R/scripts/perf_stain.py 4096 > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/rp_$c $OUT/${TAG}_stain_pmc_${c}.txt > /dev/null
done
cd $R
timeout 400 python scripts/perf_stain.py 4096 2>&1 | grep -v amdgpu > $OUT/${TAG}_perf_stain.txt
timeout 300 python scripts/perf_conv.py 1024 2>&1 | grep -v amdgpu > $OUT/${TAG}_perf_conv.txt; tail -1 $OUT/${TAG}_perf_conv.txt
timeout 400 python scripts/perf_kernels.py 2>&1 | grep stage > $OUT/${TAG}_perf_kernels.jsonl
timeout 600 python bench.py --config hovernet --steps 5 --warmup 2 > $OUT/${TAG}_bench_hovernet.json 2> /dev/null; cut -c1-300 $OUT/${TAG}_bench_hovernet.json
timeout 300 python bench.py --config vahadane --steps 5 --warmup 2 > $OUT/${TAG}_bench_vahadane.json 2> /dev/null; cut -c1-200 $OUT/${TAG}_bench_vahadane.json
timeout 600 python bench.py --config semantic --steps 1 --warmup 1 > $OUT/${TAG}_bench_semantic.json 2> /dev/null; cut -c1-300 $OUT/${TAG}_bench_semantic.json
ls $OUT | grep $TAG | wc -l
