#!/bin/bash
# round 5: outbid-skip + work hand-off + box-hierarchy seeds: parity, then ms per call on the four regimes per knob
cd "$(dirname "$0")/.."
O=gpurun_out/r5_emd1; mkdir -p $O
{
echo "== parity, defaults"; timeout 600 python tools/emd_ab.py --parity --parity32 2>&1 | grep -v amdgpu.ids | grep parity
echo "== parity, hand-off + skip forced"; SN_EMD_STEAL=2 SN_EMD_SKIP=2 timeout 600 python tools/emd_ab.py --parity --parity32 2>&1 | grep parity
echo "== parity on the regimes (B = 4, whole clouds)"; AB_BS=4 timeout 900 python tools/emd_regimes.py --parity scatter untrained 2>&1 | grep -v amdgpu.ids
echo "== parity on the regimes, forced"; SN_EMD_STEAL=2 SN_EMD_SKIP=2 AB_BS=4 timeout 900 python tools/emd_regimes.py --parity scatter untrained 2>&1 | grep -v amdgpu.ids
} > $O/parity.txt 2>&1
{
for v in "AB_LIB=tools/ab/lib_r4.so" "X=default" "SN_EMD_STEAL=0" "SN_EMD_SKIP=0" "SN_EMD_SEED=window" "SN_EMD_STEAL=2" "SN_EMD_SKIP=2" "AB_LIB=tools/ab/lib_reshare4.so" "AB_LIB=tools/ab/lib_reshare2.so" "SN_EMD_SCAN=1024" "SN_EMD_SCAN=2048"; do
  echo "== $v"; env $v timeout 600 python tools/emd_regimes.py 2>&1 | grep regime
done
} > $O/knobs.txt 2>&1
SN_EMD_DIAG=2 AB_BS=4 python tools/emd_regimes.py scatter untrained uniform > $O/phases_b4.txt 2>&1
SN_EMD_DIAG=2 AB_BS=32 python tools/emd_regimes.py scatter untrained uniform > $O/phases_b32.txt 2>&1
timeout 900 python -m pytest tests/test_emd.py -m gpu -x -q > $O/pytest_emd.txt 2>&1
tail -3 $O/pytest_emd.txt; cat $O/parity.txt; cat $O/knobs.txt
