#!/bin/bash
# A/B of the persistent Winograd form against one block per workgroup (developer switch), per layer and over the trunk.
# usage (GPU box): bash scripts/wino_persist_ab.sh [batch=4096] [patch=256]
cd "${GRAFT_REPO_ROOT:-.}"
B=${1:-4096}; P=${2:-256}
for round in 1 2; do
  echo "== persistent (default), round $round"; python scripts/perf_wino.py $B $P
  echo "== TIA_WINO_NO_PERSIST=1, round $round"; TIA_DEV=1 TIA_WINO_NO_PERSIST=1 python scripts/perf_wino.py $B $P
done
