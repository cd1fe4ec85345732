#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 1200 python -m pytest tests/test_chamfer.py tests/test_emd.py tests/test_fullsize.py tests/test_dropin.py tests/test_metrics.py tests/test_robustness.py -m gpu -q -x 2>&1 | tail -3
AB_BS=32,4 timeout 300 python tools/emd_ab.py 2>&1 | grep "ms per call"
python tools/chamfer_collapsed.py 2>&1 | grep chamfer
timeout 300 python tools/strong_share.py 2>&1 | grep -v amdgpu | cut -c1-200
