#!/bin/bash
# every kernel of one EMD forward call (iters=1) with its duration (run on the GPU box)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && EMD_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_emd1 -- python $R/tools/quick_emd.py > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_emd1/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# first call with iters=1: from the first cloud_sort/emd_init to the first calcdist
start = next(i for i, r in enumerate(rows) if "cloud_sort_count" in r["Kernel_Name"] or "emd_init" in r["Kernel_Name"])
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:start + 14]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us  {r["Kernel_Name"][:70]}')
PY
rm -rf gpurun_out/prof_emd1
