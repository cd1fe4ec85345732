#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
echo "== chamfer tests"; timeout 900 python -m pytest tests/test_chamfer.py tests/test_dropin.py tests/test_metrics.py -m gpu -q 2>&1 | tail -3
echo "== collapsed: new"; python tools/chamfer_collapsed.py 2>&1 | grep chamfer
echo "== collapsed: old"; AB_LIB=tools/ab/lib_cdold.so python tools/chamfer_collapsed.py 2>&1 | grep chamfer
