#!/bin/bash
# per-iteration emd_bid durations of one 50-iteration EMD call at C2 (run on the GPU box)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_emd -- python $R/tools/emd_probe.py > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_emd/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
bids = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "emd_bid" in r["Kernel_Name"])
d = [x[1] / 1000 for x in bids][-50:]
print("bid us per iteration:", [round(v) for v in d])
print("sum ms", sum(d) / 1000)
PY
rm -rf gpurun_out/prof_emd
