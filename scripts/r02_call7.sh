#!/bin/bash
# Round-2 GPU call 7: watershed by relaxation -- parity of the HoVer-Net post-processing, timing with and without it,
# per-kernel rocprof summary of the post-processing.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== hover tests"; timeout 900 python -m pytest tests/test_hovernet_post.py tests/test_hovernetplus.py tests/test_engine.py -m gpu -q -x 2>&1 | tail -6 | tee $OUT/r02k_pytest_hover.log
echo "== perf hover (relaxation)"; timeout 300 python scripts/perf_kernels.py hover 2>&1 | grep -v amdgpu | tee $OUT/r02k_perf_hover_relax.txt
echo "== perf hover (heap only)"; TIA_FLOOD_RELAX=0 timeout 300 python scripts/perf_kernels.py hover 2>&1 | grep -v amdgpu | tee $OUT/r02k_perf_hover_heap.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_hover; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_hover -- python $R/scripts/perf_kernels.py hover > /dev/null 2>&1
python $R/scripts/prof_summarize.py /tmp/rp_hover $OUT/r02k_hover_rocprofv3_summary.txt > /dev/null; head -30 $OUT/r02k_hover_rocprofv3_summary.txt | cut -c1-150
