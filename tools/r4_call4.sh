#!/bin/bash
# round 4, fourth GPU call: steady-state kernel tables of config 4 / 5 (after MIOpen's find phase), the auction's scan
# threshold, host enqueue time at the 8-GPU share, the new capture test.
O=gpurun_out/r4c4; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_robustness.py tests/test_harness.py tests/test_networks.py -m gpu -q 2>&1 | tail -5 ) > $O/gpu_tests_some.txt
cd /tmp
for cfg in config4 config5; do
  for st in trained_stand_in random_init; do
    rm -rf /tmp/prof_$cfg
    NS_WARMUP=3 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$cfg -o $cfg -- python $R/tools/net_step.py $cfg $st 6 2>&1 | grep "ms per step" > $R/$O/net_${cfg}_${st}.txt
    f=$(find /tmp/prof_$cfg -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python $R/tools/steady_stats.py "$f" bitwise_xor 6 40 >> $R/$O/net_${cfg}_${st}.txt 2>&1
  done
done
cd $R
{
  for sc in 128 256 384 512 1024; do echo "== SN_EMD_SCAN=$sc"; SN_EMD_SCAN=$sc AB_BS=32,16,4 timeout 300 python tools/emd_ab.py 2>&1 | grep "per call"; done
} > $O/emd_scan_threshold.txt 2>&1
( HO_B=4 timeout 300 python tools/host_overhead.py; HO_B=32 timeout 300 python tools/host_overhead.py ) > $O/host_overhead.txt 2>&1
echo done > $O/done.txt
