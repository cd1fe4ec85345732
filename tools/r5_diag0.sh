#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5_diag0
O=gpurun_out/r5_diag0
python tools/emd_regimes.py --dump $O > $O/regimes_plain.txt 2>&1
tail -8 $O/regimes_plain.txt
