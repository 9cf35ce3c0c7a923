#!/bin/bash
# First GPU call of the next round: validate and measure the float32-binning variant of stain_stats
# (tiatoolbox_amd/csrc/stain_stats.hip, -DTIA_F32_BINS=1; never run on hardware so far).
#   1. build the variant next to the product library (here, on the build host):
#        python -c "from tiatoolbox_amd import build; build.build(defines=('TIA_F32_BINS=1',), out=build.LIB_DIR/'libtiatoolbox_amd_f32bins.so')"
#      python -c "from tiatoolbox_amd import build; build.build(defines=('TIA_HEAP_CLASSIC=1',), out=build.LIB_DIR/'libtiatoolbox_amd_heapclassic.so')"
#   2. on the GPU box:
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$R/tiatoolbox_amd/lib/libtiatoolbox_amd_f32bins.so
echo "== product library"; timeout 200 python scripts/perf_stain.py 4096 2>&1 | grep -v amdgpu | head -3
echo "== f32-bins variant: parity"; TIA_LIB_PATH=$V timeout 400 python -m pytest tests/test_stain_gpu.py -m gpu -q 2>&1 | tail -4
echo "== f32-bins variant: time";   TIA_LIB_PATH=$V timeout 200 python scripts/perf_stain.py 4096 2>&1 | grep -v amdgpu | head -3
H=$R/tiatoolbox_amd/lib/libtiatoolbox_amd_heapclassic.so
echo "== skimage-style heap pop: parity + time"
TIA_LIB_PATH=$H timeout 400 python -m pytest tests/test_hovernet_post.py tests/test_tile_mode.py -m gpu -q 2>&1 | tail -3
TIA_LIB_PATH=$H timeout 200 python scripts/perf_kernels.py hover 2>&1 | grep proc_np_hv | cut -c1-150
