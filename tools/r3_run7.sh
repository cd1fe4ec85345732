#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
echo "== mds parity + timing (teams)"; timeout 900 python tools/mds_ab.py --parity 2>&1 | grep -v amdgpu | tee $O/mds_teams.txt
echo "== timing G=4"; SN_MDS_G=4 timeout 900 python tools/mds_ab.py 2>&1 | grep -v amdgpu | tee $O/mds_g4.txt
echo "== timing G=16"; SN_MDS_G=16 timeout 900 python tools/mds_ab.py 2>&1 | grep -v amdgpu | tee $O/mds_g16.txt
echo "== timing teams off"; SN_MDS_G=1 timeout 900 python tools/mds_ab.py 2>&1 | grep -v amdgpu | tee $O/mds_off.txt
echo "== mds tests"; timeout 1500 python -m pytest tests/test_mds.py tests/test_fullsize.py tests/test_dropin.py tests/test_robustness.py tests/test_harness.py -m gpu -q -k "mds or dropin or finite or step" 2>&1 | tail -4
