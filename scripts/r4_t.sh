#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "wide_tiles or (optin and X3_TILE)" 2>&1 | tail -3
