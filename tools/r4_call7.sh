#!/bin/bash
O=gpurun_out/r4c7; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
  for sc in 384 0 64; do
    echo "== scatter data, SN_EMD_SCAN=$sc, per call"; AB_DATA=scatter SN_EMD_SCAN=$sc AB_BS=32,4 timeout 300 python tools/emd_ab.py --parity 2>&1 | grep "parity\|per call"
  done
  echo "== scatter data, B=4 phases (default scan threshold)"; AB_DATA=scatter SN_EMD_DIAG=2 AB_DIAG_B=4 AB_BS=4 timeout 300 python tools/emd_ab.py 2>&1 | grep -v "amdgpu.ids"
  echo "== scatter data, B=4 phases, SN_EMD_SCAN=0"; AB_DATA=scatter SN_EMD_SCAN=0 SN_EMD_DIAG=2 AB_DIAG_B=4 AB_BS=4 timeout 300 python tools/emd_ab.py 2>&1 | grep -v "amdgpu.ids"
} > $O/emd_scatter.txt 2>&1
( timeout 600 python -m pytest tests/test_networks.py tests/test_harness.py -m gpu -q 2>&1 | tail -4 ) > $O/gpu_tests_networks.txt
for cfg in config4 config5; do
  for st in trained_stand_in random_init; do timeout 300 python tools/net_step.py $cfg $st 7 2>&1 | grep "ms per step" >> $O/net_steps.txt; done
done
cd /tmp
rm -rf /tmp/prof_c5
NS_WARMUP=3 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c5 -o c5 -- python $R/tools/net_step.py config5 trained_stand_in 6 2>&1 | grep "ms per step" > $R/$O/net_config5_steady.txt
f=$(find /tmp/prof_c5 -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $R/tools/steady_stats.py "$f" xor 6 30 >> $R/$O/net_config5_steady.txt 2>&1
cd $R
timeout 300 python tools/net_host_profile.py config5 trained_stand_in 2>&1 | head -45 > $O/host_profile_config5.txt
echo done > $O/done.txt
