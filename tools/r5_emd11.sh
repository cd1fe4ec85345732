#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5_emd11; mkdir -p $O
{
echo "== parity"; timeout 600 python tools/emd_ab.py --parity --parity32 2>&1 | grep parity
AB_BS=4 timeout 900 python tools/emd_regimes.py --parity scatter untrained surface 2>&1 | grep parity
for v in "AB_LIB=tools/ab/lib_r4.so" "X=default" "X=default" "SN_EMD_SEED=window"; do
  echo "== $v"; env $v timeout 600 python tools/emd_regimes.py 2>&1 | grep regime
done
echo "== bench r4"; AB_LIB=tools/ab/lib_r4.so python bench.py --no-network-steps --no-other-ops --no-cpu-baseline --no-literal-radii 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
echo "== bench new"; python bench.py --no-network-steps --no-other-ops --no-cpu-baseline --no-literal-radii 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
} > $O/knobs.txt 2>&1
cat $O/knobs.txt
