#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03j; mkdir -p $O
for spec in "new::" "prev:tools/ab/lib_prev.so:"; do
  name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}
  echo "-- $name"
  AB_LIB=$lib AB_BS=32,16,8,4,1 timeout 600 python tools/emd_ab.py --parity --parity32 2>&1 | grep "ms per call\|parity" | tee $O/ab_$name.txt
done
AB_LIB= AB_BS=32,4 timeout 600 python tools/emd_ab.py 2>&1 | grep "ms per call"
AB_LIB=tools/ab/lib_prev.so AB_BS=32,4 timeout 600 python tools/emd_ab.py 2>&1 | grep "ms per call"
echo "== emd tests"; timeout 1500 python -m pytest tests/test_emd.py tests/test_fullsize.py tests/test_dropin.py -m gpu -q -x -k "emd or dropin" 2>&1 | tail -3
bash tools/r3_fuzz.sh
