#!/bin/bash
# Matrix-pipe occupancy of a kernel from the hardware counters: ONE rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
# GRBM_GUI_ACTIVE and SQ_WAVE_CYCLES over a workload -> gpurun_out/TAG_STEM_pmc_MFMA.txt (means per dispatch per kernel).
#   mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)      (bench.py: pmc_mfma_busy)
#   usage: pmc_mfma.sh TAG STEM command...
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; STEM=$2; shift 2
mkdir -p $R/gpurun_out
rm -rf /tmp/rp_mfma
(cd /tmp && TMPDIR=/tmp timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d /tmp/rp_mfma -- "$@" > /tmp/rp_mfma.out 2>&1)
python $R/scripts/prof_summarize.py /tmp/rp_mfma $R/gpurun_out/${TAG}_${STEM}_pmc_MFMA.txt > /dev/null
grep -E "MFMA_BUSY|GUI_ACTIVE" $R/gpurun_out/${TAG}_${STEM}_pmc_MFMA.txt | cut -c1-130
