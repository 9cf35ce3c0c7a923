#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_semantic.py tests/test_fullsize_parity.py tests/test_engine.py -m gpu -q 2>&1 | tail -4
