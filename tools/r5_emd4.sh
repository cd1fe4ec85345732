#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5_emd4; mkdir -p $O
{
for v in "AB_LIB=tools/ab/lib_r4.so" "X=default" "AB_LIB=tools/ab/lib_nosteal.so" "SN_EMD_STEAL=0" "SN_EMD_SKIP=0" "SN_EMD_STEAL=0 SN_EMD_SKIP=0"; do
  echo "== $v"; env $v SN_EMD_DIAG=2 AB_BS=4 timeout 600 python tools/emd_regimes.py scatter untrained 2>&1 | grep -v amdgpu.ids | grep "regime\|it  0\|it  1\|it  2\|it  5\|it 10\|it 30\|sum over"
done
} > $O/bisect.txt 2>&1
cat $O/bisect.txt
