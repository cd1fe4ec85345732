#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5_emd15; mkdir -p $O
{
echo "== parity"; timeout 600 python tools/emd_ab.py --parity --parity32 2>&1 | grep parity
AB_BS=4 timeout 900 python tools/emd_regimes.py --parity scatter untrained surface 2>&1 | grep parity
for v in "AB_LIB=tools/ab/lib_r4.so" "X=default" "X=default"; do
  echo "== $v"; env $v timeout 600 python tools/emd_regimes.py 2>&1 | grep regime
done
} > $O/knobs.txt 2>&1
SN_EMD_DIAG=2 AB_BS=32 python tools/emd_regimes.py uniform > $O/phases_b32.txt 2>&1
SN_EMD_DIAG=2 AB_BS=4 python tools/emd_regimes.py uniform > $O/phases_b4.txt 2>&1
timeout 900 python -m pytest tests/test_emd.py tests/test_fullsize.py -m gpu -x -q -k emd > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
cat $O/knobs.txt; grep -v amdgpu $O/phases_b32.txt | grep "it 10\|it 20\|it 30\|sum"; grep -v amdgpu $O/phases_b4.txt | grep "it 10\|it 20\|it 30\|sum"
