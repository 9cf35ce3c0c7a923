#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_hovernet_post.py -m gpu -q -x -k "fused or nucleus_instance" 2>&1 | tail -4
echo "== bench hovernet (mfma)"; timeout 900 python bench.py --config hovernet --steps 3 --warmup 1 > $OUT/r02n_bench_hovernet.json 2> $OUT/r02n_bench_hovernet.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02n_bench_hovernet.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['roofline']['backbone']))
PY
