#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TIA_LIB_PATH=$R/tiatoolbox_amd/lib/libtiatoolbox_amd_tt.so timeout 300 python scripts/perf_hover_post.py 256 2 2>&1 | grep "stamps\|proc_np" | tail -4
