#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); O=gpurun_out/r06_f; mkdir -p $O; export TMPDIR=/tmp
echo "== mds parity"; timeout 2400 python -m pytest tests/test_mds.py tests/test_fullsize.py -m gpu -q -x -k "mds" 2>&1 | tail -4 | tee $O/tests.txt
timeout 900 python tools/mds_ab.py --parity 2>&1 | grep -v amdgpu | tee $O/mds_ab.txt
for gg in 16 8; do echo "SN_MDS_G=$gg"; SN_MDS_G=$gg timeout 900 python tools/mds_ab.py 2>&1 | grep "B=4 \|B=32 " ; done | tee -a $O/mds_ab.txt
echo "== stamps"; AB_LIB=tools/ab/lib_mdsstamps.so timeout 600 python tools/mds_dense_stamps.py 2>&1 | grep "mds dense" | tee $O/mds_dense_stamps.txt
echo "== fuzz (mds / expansion / p2i)"; timeout 400 python tools/fuzz_parity2.py 120 61 2>&1 | grep -v amdgpu | tail -3 | tee $O/fuzz2.txt
echo done > $O/done.txt
