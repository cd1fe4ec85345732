#!/bin/bash
# usage: tools/pmc_kernel.sh <kernel-substring> <python script...>   (run on the GPU box)
# collects two PMC passes (counters only: --kernel-trace + --pmc) and prints per-dispatch values
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; shift
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc1 -- python $R/$@ > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc2 -- python $R/$@ > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob
for d in ("pmc1", "pmc2"):
    fs = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no counter file"); continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        if "$K" not in r["Kernel_Name"]:
            continue
        agg.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(agg)
    for did in ids[:2]:
        print(d, "dispatch", did, {k: f"{v:.4g}" for k, v in agg[did].items()})
    tail = ids[-20:]   # e.g. the late auction iterations
    keys = sorted(agg[tail[0]])
    print(d, f"mean of the last {len(tail)} dispatches", {k: f"{sum(agg[i].get(k, 0.0) for i in tail) / len(tail):.4g}" for k in keys})
PY
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
