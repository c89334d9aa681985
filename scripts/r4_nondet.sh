#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; cd "$R"
GAST_TEST_H16=f16 timeout 600 python scripts/r4_nondet.py big_dgrad_gather_bwd bf16 2>&1 | tail -8
timeout 600 python scripts/r4_nondet.py big_dgrad_gather_bwd bf16 2>&1 | tail -4
timeout 600 python scripts/r4_nondet.py big_dgrad_gather_bwd f32 2>&1 | tail -4
