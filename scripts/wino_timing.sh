#!/bin/bash
# Phase timing of the Winograd kernel on a library built with -DTIA_WINO_TIMING=1 (build it first, here or in the container:
#   python -c "from tiatoolbox_amd import build as b; b.build(defines=('TIA_WINO_TIMING=1',), out=b.LIB_DIR / 'libtiatoolbox_amd_winotiming.so')").
# Prints the phase lines of two waves of one workgroup per layer shape, persistent form and (developer switch) one block per workgroup.
# usage (GPU box): bash scripts/wino_timing.sh [batch=1024] [patch=256]
cd "${GRAFT_REPO_ROOT:-.}"
B=${1:-1024}; P=${2:-256}
export TIA_LIB_PATH=$PWD/tiatoolbox_amd/lib/libtiatoolbox_amd_winotiming.so
echo "== persistent (default)"; python scripts/perf_wino.py $B $P 2>&1 | grep "wino wg" | awk '{k=$5 $6 $7; c[k]++; if (c[k] <= 2) print}'
echo "== TIA_WINO_NO_PERSIST=1"; TIA_DEV=1 TIA_WINO_NO_PERSIST=1 python scripts/perf_wino.py $B $P 2>&1 | grep "wino wg" | awk '{k=$5 $6 $7; c[k]++; if (c[k] <= 2) print}'
