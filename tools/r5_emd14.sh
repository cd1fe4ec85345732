#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5_emd14; mkdir -p $O
timeout 1500 python -m pytest tests/test_emd.py tests/test_fullsize.py tests/test_p2i.py -m gpu -x -q > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
for v in "AB_LIB=tools/ab/lib_r4.so" "X=default" "X=default"; do
  echo "== $v"; env $v timeout 600 python tools/emd_regimes.py 2>&1 | grep regime
done > $O/regimes.txt 2>&1
cat $O/regimes.txt
