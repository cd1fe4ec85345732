#!/bin/bash
# HBM traffic of emd_bid_kernel from PMC counters, separate passes (run on the GPU box)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcF -- python $R/tools/emd_probe.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcW -- python $R/tools/emd_probe.py > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob
out = {}
for d, name in (("pmcF", "FETCH_SIZE"), ("pmcW", "WRITE_SIZE")):
    fs = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0]))
            if "emd_bid_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    out[name] = (len(vals), sum(vals) / max(len(vals), 1), max(vals), min(vals))
    print(name, "launches", len(vals), "avg KB/launch %.1f" % out[name][1], "max %.1f" % out[name][2], "min %.1f" % out[name][3])
PY
rm -rf gpurun_out/pmcF gpurun_out/pmcW
