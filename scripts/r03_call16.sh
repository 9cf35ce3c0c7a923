#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_thin_head_gpu.py tests/test_semantic.py tests/test_hovernetplus.py tests/test_hovernet_post.py tests/test_tile_mode.py tests/test_engine.py -m gpu -q 2>&1 | tail -15
