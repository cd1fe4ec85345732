#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for v in 0 8 16; do echo "== BENCH_EMD_FIRST_MAX=$v"; BENCH_EMD_FIRST_MAX=$v timeout 600 python tools/strong_share.py 2>&1 | grep -v amdgpu | cut -c1-90; done
