#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python scripts/prof_hovernet_host.py 2>&1 | grep -v "amdgpu\|No local" | head -64 > $R/gpurun_out/r03v_prof_hovernet_host.txt; head -60 $R/gpurun_out/r03v_prof_hovernet_host.txt | cut -c1-150
