#!/bin/bash
# Round-2 GPU call 8: where does the HoVer-Net / UNet forward pass spend its time? (kernel trace of bench configs 3 and 4)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in hovernet semantic; do
  rm -rf /tmp/rp_$cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$cfg -- python $R/bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline > $OUT/r02j_bench_${cfg}_under_rocprof.json 2> $OUT/r02j_bench_${cfg}.err
  python $R/scripts/prof_summarize.py /tmp/rp_$cfg $OUT/r02j_${cfg}_rocprofv3_summary.txt > /dev/null
  echo "== $cfg"; head -28 $OUT/r02j_${cfg}_rocprofv3_summary.txt | cut -c1-170
done
