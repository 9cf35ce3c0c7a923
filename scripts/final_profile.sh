#!/bin/bash
# Round-end measurement pass on the GPU box: smoke, bench (+cpu_baseline), rocprofv3 kernel trace of the bench,
# separate --pmc passes for HBM traffic of the stain kernels, and the stage-level measurements of the other
# kernel families with their kernel traces.  Everything lands in gpurun_out/ (copied to profiles/ afterwards).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${1:-r01c}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
cd /tmp
rm -rf /tmp/rp_bench; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_bench -- \
    python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
python $R/scripts/prof_summarize.py /tmp/rp_bench $OUT/${TAG}_bench_rocprofv3_summary.txt > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_$c; timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/rp_$c -- \
      python $R/scripts/perf_stain.py 4096 > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/rp_$c $OUT/${TAG}_stain_pmc_${c}.txt > /dev/null
done
cd $R
timeout 400 python scripts/perf_stain.py 4096 2>&1 | grep -v amdgpu > $OUT/${TAG}_perf_stain.txt
timeout 400 python scripts/perf_kernels.py 2>&1 | grep stage > $OUT/${TAG}_perf_kernels.jsonl
cd /tmp
for s in reinhard mask hover; do
  rm -rf /tmp/rp_$s; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$s -- \
      python $R/scripts/perf_kernels.py $s > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/rp_$s $OUT/${TAG}_perf_${s}_rocprofv3_summary.txt > /dev/null
done
ls -la $OUT | tail -20
