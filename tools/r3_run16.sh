#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python tools/strong_share.py 2>&1 | grep -v amdgpu | cut -c1-100
