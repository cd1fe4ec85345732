#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5_emd2; mkdir -p $O
{
for v in "AB_LIB=tools/ab/lib_r4.so" "X=default" "SN_EMD_STEAL=-1" "AB_LIB=tools/ab/lib_nosteal.so" "AB_LIB=tools/ab/lib_noclock.so" "SN_EMD_STEAL=0"; do
  echo "== $v"; env $v SN_EMD_DIAG=2 AB_BS=32 timeout 600 python tools/emd_regimes.py uniform 2>&1 | grep -v amdgpu.ids | grep "regime\|it  0\|it  3\|it  5\|it 10\|it 20\|it 30\|sum over"
done
} > $O/bisect.txt 2>&1
cat $O/bisect.txt
