#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $R/gpurun_out/r03p_bench.json 2> $R/gpurun_out/r03p_bench.err; echo "bench rc=$?"
python - <<'PY'
import json,os
d=json.loads(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r03p_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], r["kernel"], r["achieved"], r["frac"], r["launches_per_step"], r["traffic"])
for k,v in r["other_kernels"].items(): print(k, v.get("achieved"), v.get("frac"), v.get("launches_per_step"))
PY
