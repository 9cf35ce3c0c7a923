#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_semantic.py -m gpu -q -x 2>&1 | tail -4
for be in mfma miopen; do
echo "== bench semantic ($be)"; timeout 900 python bench.py --config semantic --steps 1 --warmup 1 --conv-backend $be --no-cpu-baseline > $OUT/r02o_bench_semantic_$be.json 2> $OUT/r02o_bench_semantic_$be.err; BE=$be python - <<'PY'
import json, os
d=json.loads(open('gpurun_out/r02o_bench_semantic_%s.json' % os.environ['BE']).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['roofline']['backbone']))
PY
tail -2 $OUT/r02o_bench_semantic_$be.err
done
