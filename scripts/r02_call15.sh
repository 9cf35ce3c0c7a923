#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_hovernet_post.py tests/test_engine.py -m gpu -q -x -k "fused or nucleus_instance or mfma" 2>&1 | tail -3
echo "== fwd hovernet"; timeout 300 python scripts/perf_hovernet_fwd.py 32 2>&1 | grep forward
echo "== perf_conv"; timeout 300 python scripts/perf_conv.py 1024 2>&1 | grep total | sed "s/| miopen.*//"
