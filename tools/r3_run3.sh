#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
echo "== parity + A/B"
for spec in "new::" "prev:tools/ab/lib_prev.so:"; do
  name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; envs=${rest#*:}
  echo "-- $name"
  env $envs AB_LIB=$lib AB_BS=32,16,8,4,1 timeout 600 python tools/emd_ab.py --parity --parity32 2>&1 | grep "ms per call\|parity" | tee $O/ab_$name.txt
done
echo "== emd tests"; timeout 1500 python -m pytest tests/test_emd.py tests/test_fullsize.py tests/test_dropin.py -m gpu -q -x -k "emd or dropin" 2>&1 | tail -5
echo "== network tests"; timeout 1500 python -m pytest tests/test_networks.py -m gpu -q -s 2>&1 | grep -v Warning | tail -12
echo "== phases"
for bb in 32 4; do
  SN_EMD_DIAG=2 AB_BS=$bb AB_DIAG_B=$bb timeout 600 python tools/emd_ab.py > $O/phases_b$bb.txt 2>&1; tail -3 $O/phases_b$bb.txt
done
echo "== bid stamps"
for bb in 32 4; do
  BID_STAMPS=1 SN_EMD_DIAG=1 AB_LIB=tools/ab/lib_stamps.so AB_BS=$bb AB_DIAG_B=$bb timeout 600 python tools/emd_ab.py 2>&1 | tail -19 | tee $O/stamps_b$bb.txt
done
