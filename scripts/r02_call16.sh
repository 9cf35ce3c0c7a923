#!/bin/bash
# Round-2 GPU call 16: final state -- full GPU suite, default bench, hovernet bench (after the 1x1 tile rule and the second epilogue output)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r02q
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/${TAG}_pytest_gpu.log; cat $OUT/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-260 $OUT/${TAG}_bench.json
timeout 600 python bench.py --config hovernet --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_hovernet.json 2> /dev/null; cut -c1-260 $OUT/${TAG}_bench_hovernet.json
