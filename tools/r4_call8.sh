#!/bin/bash
O=gpurun_out/r4c8; mkdir -p $O
export TMPDIR=/tmp
{
  for lib in "" tools/ab/lib_emdhead.so; do
    echo "== lib ${lib:-cost-weighted split}: uniform"; AB_LIB=$lib AB_BS=32,16,8,4 timeout 300 python tools/emd_ab.py --parity --parity32 2>&1 | grep "parity\|per call"
    echo "== lib ${lib:-cost-weighted split}: scatter"; AB_LIB=$lib AB_DATA=scatter AB_BS=32,4 timeout 300 python tools/emd_ab.py --parity 2>&1 | grep "parity\|per call"
    echo "== lib ${lib:-cost-weighted split}: surface"; AB_LIB=$lib AB_DATA=surface AB_BS=32,4 timeout 300 python tools/emd_ab.py 2>&1 | grep "per call"
  done
  echo "== cost-weighted split, SN_EMD_COSTW=0: uniform / scatter"; SN_EMD_COSTW=0 AB_BS=32,4 timeout 300 python tools/emd_ab.py 2>&1 | grep "per call"; SN_EMD_COSTW=0 AB_DATA=scatter AB_BS=32,4 timeout 300 python tools/emd_ab.py 2>&1 | grep "per call"
  echo "== scatter data, B=4 phases (cost-weighted)"; AB_DATA=scatter SN_EMD_DIAG=2 AB_DIAG_B=4 AB_BS=4 timeout 300 python tools/emd_ab.py 2>&1 | grep "tail mean\|it  0\|it  5\|it 20\|per call"
} > $O/emd_costw.txt 2>&1
( timeout 600 python -m pytest tests/test_emd.py tests/test_fullsize.py -m gpu -q -k "emd" 2>&1 | tail -4 ) > $O/gpu_tests_emd.txt
for cfg in config4 config5; do for st in trained_stand_in random_init; do timeout 300 python tools/net_step.py $cfg $st 7 2>&1 | grep "ms per step" >> $O/net_steps.txt; done; done
echo done > $O/done.txt
