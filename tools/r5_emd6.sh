#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5_emd6; mkdir -p $O
{
for m0 in 0 8 16 24; do
  echo "== M0=$m0"; SN_EMD_DIAG_M0=$m0 SN_EMD_DIAG=2 AB_BS=4 timeout 600 python tools/emd_regimes.py scatter 2>&1 | grep -v amdgpu.ids | grep "regime\|it  0\|it  1\|it  2\|it  5\|it 10\|it 30"
done
} > $O/allwgs.txt 2>&1
cat $O/allwgs.txt
