#!/bin/bash
# full GPU suite + smoke + default bench + semantic / hovernet configs after the stats / half-conv / UNet-stem changes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03g
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $OUT/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/${TAG}_smoke.log
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 3000 $OUT/${TAG}_bench.json
timeout 600 python bench.py --config semantic > $OUT/${TAG}_bench_semantic.json 2>> $OUT/${TAG}_bench.err; tail -c 1500 $OUT/${TAG}_bench_semantic.json
timeout 600 python bench.py --config hovernet > $OUT/${TAG}_bench_hovernet.json 2>> $OUT/${TAG}_bench.err; tail -c 1500 $OUT/${TAG}_bench_hovernet.json
tail -5 $OUT/${TAG}_bench.err
