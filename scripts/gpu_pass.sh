#!/bin/bash
# Generic GPU-box pass of round 4: GPU test suite, smoke, environment probe, then whatever perf scripts are named.
#   usage: gpu_pass.sh TAG [cmd ...]   (each extra argument is run through bash -c with its output in gpurun_out/TAG_<n>.txt)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${1:-r04a}; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
python scripts/probe_env.py > $OUT/${TAG}_env_probe.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $OUT/${TAG}_pytest_gpu.log; tail -8 $OUT/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/${TAG}_smoke.log
i=0
for c in "$@"; do
  i=$((i+1))
  timeout 900 bash -c "$c" > $OUT/${TAG}_$i.txt 2>&1; echo "[$i] rc=$? : $c"; grep -v "amdgpu.ids\|No local weights" $OUT/${TAG}_$i.txt | tail -${TAIL:-12}
done
