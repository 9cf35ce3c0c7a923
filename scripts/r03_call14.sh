#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=r03h
cd $R
timeout 600 python -m pytest tests/test_semantic.py tests/test_stem_gpu.py tests/test_hovernet_post.py -m gpu -q 2>&1 | tail -5
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_semantic.py --deselect tests/test_stem_gpu.py --deselect tests/test_hovernet_post.py 2>&1 | tail -5 | tee $OUT/${TAG}_pytest_gpu_rest.log
timeout 300 python scripts/perf_hovernet_layers.py hovernet 32 > $OUT/${TAG}_hovernet_layers.txt 2>&1; head -40 $OUT/${TAG}_hovernet_layers.txt
timeout 300 python scripts/perf_hovernet_layers.py unet 8 > $OUT/${TAG}_unet_layers.txt 2>&1; head -40 $OUT/${TAG}_unet_layers.txt
timeout 600 python bench.py --config hovernet > $OUT/${TAG}_bench_hovernet.json 2>> $OUT/${TAG}_bench.err; python - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r03h_bench_hovernet.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"].get("postproc_incl_tables_ms"))
PY
