#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03k
cd $R
scripts/bin/mfma_clock 20000 1 | tee $OUT/${TAG}_mfma_clock.txt
scripts/bin/mfma_clock 20000 2 | tee -a $OUT/${TAG}_mfma_clock.txt
scripts/bin/mfma_clock 20000 4 | tee -a $OUT/${TAG}_mfma_clock.txt
timeout 600 python bench.py --config semantic > $OUT/${TAG}_bench_semantic.json 2>> $OUT/${TAG}_bench.err
timeout 600 python bench.py --config hovernet > $OUT/${TAG}_bench_hovernet.json 2>> $OUT/${TAG}_bench.err
python - <<'PY'
import json,os
for n in ("semantic","hovernet"):
    d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+f"/gpurun_out/r03k_bench_{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["unit"], d["roofline"]["backbone"])
PY
