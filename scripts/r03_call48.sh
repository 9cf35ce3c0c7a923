#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python bench.py --config vahadane --steps 5 --warmup 2 > $R/gpurun_out/r03p_bench_vahadane.json 2> $R/gpurun_out/r03p_bench_vahadane.err; echo rc=$?; tail -3 $R/gpurun_out/r03p_bench_vahadane.err
python - <<'PY'
import json,os
e=json.loads(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r03p_bench_vahadane.json").read().strip().splitlines()[-1])
print(e["value"], e.get("extras"))
PY
