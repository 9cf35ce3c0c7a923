#!/bin/bash
# Round-6 measurement pass on the GPU box (everything lands in gpurun_out/, copied to profiles/ afterwards): environment probe, full GPU
# test suite, smoke, separate --pmc passes (FETCH_SIZE / WRITE_SIZE: the direct trunk convolutions at 256^2 and 224^2, the Winograd layers,
# the stain kernels at the headline size, the classic stages of bench_classic.py; SQ_VALU_MFMA_BUSY_CYCLES for both convolution kernels),
# the default bench (value = conv_algo "auto"; extras.cnn_direct, extras.classic, extras.configs, cpu_baseline), rocprofv3 kernel trace
# of the bench, per-layer tables (direct vs Winograd at both sizes, persistent Winograd form vs one block per workgroup), the stain /
# Reinhard / large-image tables, the other bench configurations.  The Vahadane / HoVer-Net post-processing / canvas PMC passes are not
# repeated (code unchanged since profiles/r05x_*: bench_configs reads those).
#   usage: final_profile_r06.sh TAG COMMIT [skip_tests]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${1:-r06z}
COMMIT=${2:-unknown}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
echo "$COMMIT" > $OUT/${TAG}_COMMIT.txt
python scripts/probe_env.py > $OUT/${TAG}_env_probe.txt 2>&1
if [ "${3:-}" != "skip_tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/${TAG}_pytest_gpu.log; cat $OUT/${TAG}_pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"
fi
pmc() {  # pmc STEM COMMAND...: one pass per counter, summaries named ${TAG}_${STEM}_pmc_${COUNTER}.txt
  local stem=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/rp_$stem$c; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/rp_$stem$c -- "$@" > /tmp/rp_$stem$c.out 2>&1)
    python $R/scripts/prof_summarize.py /tmp/rp_$stem$c $OUT/${TAG}_${stem}_pmc_${c}.txt > /dev/null
    # a workload that ran a known number of forwards says so ("PMC forwards=N"): bytes per layer CALL = total / (launches per forward x N)
    grep -h "^PMC forwards=" /tmp/rp_$stem$c.out | head -1 | sed 's/^/# /' >> $OUT/${TAG}_${stem}_pmc_${c}.txt
  done
}
pmc trunk4096 python $R/scripts/perf_trunk.py 4096 256 5 pmc
pmc trunk224_4096 python $R/scripts/perf_trunk.py 4096 224 5 pmc
pmc wino4096 python $R/scripts/perf_wino.py 4096 256 pmc
pmc stain python $R/scripts/perf_stain.py 4096 256
for st in reinhard mask luminosity augment; do pmc classic_$st python $R/bench_classic.py --pmc $st --calls 2; done
bash $R/scripts/pmc_mfma.sh $TAG wino4096 python $R/scripts/perf_wino.py 4096 256 pmc
bash $R/scripts/pmc_mfma.sh $TAG trunk4096 python $R/scripts/perf_trunk.py 4096 256 5 pmc
grep -h "conv3x3\|conv_mfma\|stem7x7" $OUT/${TAG}_trunk4096_pmc_*SIZE.txt $OUT/${TAG}_wino4096_pmc_*SIZE.txt | cut -c1-130
# the benches read the traffic of their kernels from profiles/: make this pass visible to the runs below
cp $OUT/${TAG}_*_pmc_*.txt $OUT/${TAG}_COMMIT.txt $R/profiles/ 2>/dev/null
cd $R
( time timeout 1200 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err ) 2> $OUT/${TAG}_bench.time; echo "bench rc=$?"; cat $OUT/${TAG}_bench.time | tr '\n' ' '; echo; cut -c1-300 $OUT/${TAG}_bench.json
(cd /tmp && rm -rf /tmp/rp_bench; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_bench -- \
    python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null)
python $R/scripts/prof_summarize.py /tmp/rp_bench $OUT/${TAG}_bench_rocprofv3_summary.txt > /dev/null; head -14 $OUT/${TAG}_bench_rocprofv3_summary.txt | cut -c1-150
timeout 400 python bench_classic.py > $OUT/${TAG}_bench_classic.txt 2> /dev/null; grep -c launch_ms $OUT/${TAG}_bench_classic.txt
for n in 4096 1024; do for hw in 256 224; do timeout 300 python scripts/perf_trunk.py $n $hw 2>&1 | grep -v "amdgpu\|No local"; done; done > $OUT/${TAG}_perf_trunk.txt; cat $OUT/${TAG}_perf_trunk.txt
timeout 300 python scripts/perf_wino.py 4096 256 2>&1 | grep -v "amdgpu\|No local" > $OUT/${TAG}_perf_wino256.txt; tail -1 $OUT/${TAG}_perf_wino256.txt
timeout 300 python scripts/perf_wino.py 4096 224 2>&1 | grep -v "amdgpu\|No local" > $OUT/${TAG}_perf_wino224.txt; tail -1 $OUT/${TAG}_perf_wino224.txt
timeout 600 bash scripts/wino_persist_ab.sh 4096 256 2>&1 | grep -v "amdgpu\|No local" > $OUT/${TAG}_wino_persist_ab_4096.txt; grep "^13\|^==" $OUT/${TAG}_wino_persist_ab_4096.txt
timeout 400 python scripts/perf_stain.py 4096 256 2>&1 | grep -v amdgpu > $OUT/${TAG}_perf_stain.txt; grep "^stats\|^apply" $OUT/${TAG}_perf_stain.txt
timeout 300 python scripts/perf_big_image.py 2>&1 | grep -v amdgpu > $OUT/${TAG}_perf_big_image.txt; tail -6 $OUT/${TAG}_perf_big_image.txt
timeout 300 python scripts/perf_reinhard_sizes.py 2>&1 | grep -v amdgpu > $OUT/${TAG}_perf_reinhard_sizes.txt; tail -4 $OUT/${TAG}_perf_reinhard_sizes.txt
timeout 300 python scripts/perf_hovernet_layers.py hovernet 32 2>&1 | grep -v "amdgpu\|No local" > $OUT/${TAG}_hovernet_layers.txt; head -1 $OUT/${TAG}_hovernet_layers.txt
timeout 300 python scripts/perf_hovernet_layers.py unet 8 2>&1 | grep -v "amdgpu\|No local" > $OUT/${TAG}_unet_layers.txt; head -1 $OUT/${TAG}_unet_layers.txt
timeout 600 python bench.py --config hovernet --steps 5 --warmup 2 > $OUT/${TAG}_bench_hovernet.json 2> /dev/null; cut -c1-200 $OUT/${TAG}_bench_hovernet.json
timeout 600 python bench.py --config vahadane --steps 5 --warmup 2 > $OUT/${TAG}_bench_vahadane.json 2> /dev/null; cut -c1-200 $OUT/${TAG}_bench_vahadane.json
timeout 600 python bench.py --config semantic --steps 1 --warmup 1 > $OUT/${TAG}_bench_semantic.json 2> /dev/null; cut -c1-200 $OUT/${TAG}_bench_semantic.json
ls $OUT | grep $TAG | wc -l
