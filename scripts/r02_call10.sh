#!/bin/bash
# Round-2 GPU call 10: fused HoVer-Net forward (MFMA convolutions) -- parity, then the hovernet bench with both backends.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_hovernet_post.py tests/test_hovernetplus.py -m gpu -q -x 2>&1 | tail -8 | tee $OUT/r02m_pytest_hover.log
echo "== bench hovernet (mfma)"; timeout 900 python bench.py --config hovernet --steps 3 --warmup 1 > $OUT/r02m_bench_hovernet.json 2> $OUT/r02m_bench_hovernet.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02m_bench_hovernet.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['roofline']['backbone']))
PY
tail -3 $OUT/r02m_bench_hovernet.err
echo "== bench hovernet (miopen)"; timeout 900 python bench.py --config hovernet --steps 3 --warmup 1 --conv-backend miopen --no-cpu-baseline > $OUT/r02m_bench_hovernet_miopen.json 2> $OUT/r02m_bench_hovernet_miopen.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02m_bench_hovernet_miopen.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['roofline']['backbone']))
PY
