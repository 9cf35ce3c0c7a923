#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_semantic.py tests/test_hovernet_post.py -m gpu -q -k "fused or semantic_segmentor" 2>&1 | tail -2
timeout 600 python bench.py --config semantic --steps 1 --warmup 1 --no-cpu-baseline > $OUT/r02r_bench_semantic.json 2> /dev/null; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02r_bench_semantic.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['backbone']['ms_per_patch'], d['roofline']['backbone']['achieved'])
PY
