#!/bin/bash
# Round-2 GPU call 1: full GPU test suite (incl. forced-tie watershed + stage planes), the two compile-time variants
# prepared last round (f32 binning for stain_stats; heapq-style pop for the perf delta of the skimage pop), new bench.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/r02a_pytest_gpu.log
echo "== product: perf_stain"; timeout 200 python scripts/perf_stain.py 4096 2>&1 | grep -v amdgpu | tee $OUT/r02a_perf_stain.txt | head -4
V=$R/tiatoolbox_amd/lib/libtiatoolbox_amd_f32bins.so
echo "== f32-bins variant: parity"; TIA_LIB_PATH=$V timeout 400 python -m pytest tests/test_stain_gpu.py -m gpu -q 2>&1 | tail -4 | tee $OUT/r02a_f32bins_pytest.log
echo "== f32-bins variant: time";   TIA_LIB_PATH=$V timeout 200 python scripts/perf_stain.py 4096 2>&1 | grep -v amdgpu | tee $OUT/r02a_f32bins_perf_stain.txt | head -4
echo "== hover perf: skimage pop (product)"; timeout 200 python scripts/perf_kernels.py hover 2>&1 | grep stage | tee $OUT/r02a_hover_classic.jsonl | cut -c1-200
H=$R/tiatoolbox_amd/lib/libtiatoolbox_amd_heapq.so
echo "== hover perf: heapq-style pop (old default)"; TIA_LIB_PATH=$H timeout 200 python scripts/perf_kernels.py hover 2>&1 | grep stage | tee $OUT/r02a_hover_heapq.jsonl | cut -c1-200
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/r02a_bench.json 2> $OUT/r02a_bench.err; echo "bench rc=$?"; tail -c 3000 $OUT/r02a_bench.json; tail -5 $OUT/r02a_bench.err
