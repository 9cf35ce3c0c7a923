#!/bin/bash
# Round-2 GPU call 3: full GPU suite (WSI mode, gather kernel, Vahadane, HoVerNet+), then the three config benches.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -30 | tee $OUT/r02c_pytest_gpu.log
for cfg in vahadane hovernet semantic; do
  echo "== bench --config $cfg"
  extra=""; [ $cfg = semantic ] && extra="--steps 2 --warmup 1"; [ $cfg = hovernet ] && extra="--steps 5 --warmup 2"; [ $cfg = vahadane ] && extra="--steps 5 --warmup 2"
  timeout 900 python bench.py --config $cfg $extra > $OUT/r02c_bench_$cfg.json 2> $OUT/r02c_bench_$cfg.err; echo "rc=$?"
  tail -c 2500 $OUT/r02c_bench_$cfg.json; grep -v amdgpu $OUT/r02c_bench_$cfg.err | tail -8
done
