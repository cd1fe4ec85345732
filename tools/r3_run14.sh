#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for spec in "base:" "bskip:tools/ab/lib_bskip.so"; do
  name=${spec%%:*}; lib=${spec#*:}
  echo "-- $name"
  AB_LIB=$lib AB_BS=32,16,4 timeout 600 python tools/emd_ab.py $( [ $rep = 1 ] && echo --parity32 ) 2>&1 | grep "ms per call\|parity"
done; done
SN_EMD_DIAG=2 AB_LIB=tools/ab/lib_bskip.so AB_BS=32 AB_DIAG_B=32 timeout 600 python tools/emd_ab.py 2>&1 | tail -2 | cut -c1-200
