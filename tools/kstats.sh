#!/bin/bash
# usage: tools/kstats.sh <python script> : rocprofv3 kernel stats top-12 (run on the GPU box)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst -- python $R/$1 2>&1 | grep -v "^W2\|rocprof" | tail -2
cd $R
f=$(find gpurun_out/kst -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.reader(open("$f")))
print(f"{'calls':>6} {'total_ms':>9} {'avg_us':>9} {'%':>6}  name")
for r in rows[1:int(__import__("os").environ.get("KTOP", "13"))]:
    print(f"{r[1]:>6} {float(r[2])/1e6:9.3f} {float(r[3])/1e3:9.1f} {float(r[4]):6.2f}  {r[0][:90]}")
PY
rm -rf gpurun_out/kst
