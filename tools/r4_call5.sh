#!/bin/bash
O=gpurun_out/r4c5; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for cfg in config4 config5; do
  for st in trained_stand_in; do
    rm -rf /tmp/prof_$cfg
    NS_WARMUP=3 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$cfg -o $cfg -- python $R/tools/net_step.py $cfg $st 6 2>&1 | grep "ms per step" > $R/$O/net_${cfg}_${st}.txt
    f=$(find /tmp/prof_$cfg -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python $R/tools/steady_stats.py "$f" xor 6 45 >> $R/$O/net_${cfg}_${st}.txt 2>&1
  done
done
cd $R
AB_BS=32,4 timeout 300 python tools/emd_ab.py --parity --parity32 2>&1 | grep "parity\|per call" > $O/emd_scan384.txt
echo done > $O/done.txt
