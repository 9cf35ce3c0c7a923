#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over the classic stages (bench_classic.py --pmc <stage>): gpurun_out/TAG_classic_<stage>_pmc_*.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
for st in reinhard mask luminosity augment; do
  timeout 300 bash $R/scripts/pmc_one.sh $TAG classic_$st python $R/bench_classic.py --pmc $st --calls 2 | grep -v "^$" | cut -c1-110
done
