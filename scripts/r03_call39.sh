#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python scripts/prof_semantic_host.py 2>&1 | grep -v "amdgpu\|No local" | head -60 > $R/gpurun_out/r03v_prof_semantic_host.txt; head -56 $R/gpurun_out/r03v_prof_semantic_host.txt | cut -c1-150
