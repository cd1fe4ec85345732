#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5_emd9; mkdir -p $O
{
echo "== parity"; timeout 600 python tools/emd_ab.py --parity --parity32 2>&1 | grep parity
AB_BS=4 timeout 900 python tools/emd_regimes.py --parity scatter untrained surface 2>&1 | grep parity
AB_LIB=tools/ab/lib_heavy8.so AB_BS=4 timeout 900 python tools/emd_regimes.py --parity scatter untrained 2>&1 | grep parity
for v in "X=default" "AB_LIB=tools/ab/lib_heavy65.so" "AB_LIB=tools/ab/lib_heavy8.so" "AB_LIB=tools/ab/lib_heavy20.so"; do
  echo "== $v"; env $v timeout 600 python tools/emd_regimes.py 2>&1 | grep regime
done
} > $O/knobs.txt 2>&1
SN_EMD_DIAG=2 AB_BS=4 python tools/emd_regimes.py scatter untrained uniform > $O/phases_b4.txt 2>&1
SN_EMD_DIAG=2 AB_BS=32 python tools/emd_regimes.py scatter untrained uniform > $O/phases_b32.txt 2>&1
cat $O/knobs.txt; grep -A12 "regime scatter" $O/phases_b4.txt | grep -v amdgpu
