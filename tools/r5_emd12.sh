#!/bin/bash
cd "$(dirname "$0")/.."
R=$GRAFT_REPO_ROOT
O=gpurun_out/r5_emd12; mkdir -p $O
export TMPDIR=/tmp
export AB_BS=32
for v in r4 new; do
  if [ $v = r4 ]; then export AB_LIB=$R/tools/ab/lib_r4.so; else unset AB_LIB; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $R/tools/emd_regimes.py uniform > /tmp/log_$v.txt 2>&1)
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v $f"; grep regime /tmp/log_$v.txt; python - <<PY
import csv
rows = list(csv.reader(open("$f")))
for r in rows[1:12]:
    print(f"{r[1]:>6} {float(r[2])/1e6:9.3f} {float(r[3])/1e3:9.1f} {float(r[4]):6.2f}  {r[0][:80]}")
PY
done > $O/kstats.txt 2>&1
cat $O/kstats.txt
