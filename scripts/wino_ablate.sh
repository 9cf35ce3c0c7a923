#!/bin/bash
# Ablations of the Winograd kernel on a library built with -DTIA_WINO_ABLATE=1
#   python -c "from tiatoolbox_amd import build as b; b.build(defines=('TIA_WINO_ABLATE=1',), out=b.LIB_DIR / 'libtiatoolbox_amd_winoablate.so')"
# bits of TIA_WINO_ABL: 1 no weight DMA in the loop, 2 no patch DMA, 64 no MFMAs, 128 nothing after the column transform, 256 no exchange, 512 no read-out
# (results are wrong by construction; the "max rel diff" column shows it).  usage (GPU box): bash scripts/wino_ablate.sh [batch=1024] [patch=256] [bits ...]
cd "${GRAFT_REPO_ROOT:-.}"
B=${1:-1024}; P=${2:-256}; shift 2
export TIA_LIB_PATH=$PWD/tiatoolbox_amd/lib/libtiatoolbox_amd_winoablate.so TIA_DEV=1
for abl in "$@"; do
  echo "== TIA_WINO_ABL=$abl"; TIA_WINO_ABL=$abl python scripts/perf_wino.py $B $P 2>&1 | grep "^3x3\|^13" | sed -e 's/direct.*| winograd/winograd/' -e 's/TF.s effective.*//'
done
