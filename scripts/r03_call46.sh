#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 300 python scripts/prof_hover_postproc_steps.py 2>&1 | grep -v amdgpu | tee $R/gpurun_out/r03v_prof_hover_postproc_steps.txt
