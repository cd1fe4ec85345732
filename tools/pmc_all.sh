#!/bin/bash
# usage (on the GPU box): tools/pmc_all.sh <out.json> [probe script, default tools/step_probe.py]
# Hardware counters of every hot kernel: four rocprofv3 passes (counters only: --kernel-trace + --pmc,
# never combined with other trace domains), per-kernel means written to <out.json>.
#   pass A: wave / VALU / wait counters       pass B: LDS, VMEM, MFMA, any-instruction activity
#   pass C: FETCH_SIZE (+ GRBM_GUI_ACTIVE)    pass D: WRITE_SIZE
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
PROBE=${1:-tools/step_probe.py}
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmcA -- python $R/$PROBE > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmcB -- python $R/$PROBE > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcC -- python $R/$PROBE > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcD -- python $R/$PROBE > /dev/null 2>&1
cd $R
python - "$OUT" <<'PY'
import csv, glob, json, sys
kernels = ["emd_auction_kernel", "emd_seed_kernel", "emd_init_kernel", "cloud_sort_count_kernel",
           "nn_search_kernel", "chamfer_bwd_lists_kernel", "chamfer_bwd_gather_kernel", "expansion_fwd_kernel",
           "p2i_gather_max_kernel", "p2i_max_bwd_accum_kernel", "p2i_bin_grouped_kernel", "p2i_absmax_kernel",
           "depth_project_views_kernel", "mds_clustered_kernel", "mds_dense_team_kernel"]
res = {k: {} for k in kernels}
for d in ("pmcA", "pmcB", "pmcC", "pmcD"):
    fs = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        res.setdefault("_missing", []).append(d)
        continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        for k in kernels:
            if k in r["Kernel_Name"]:
                a = agg.setdefault((k, r["Counter_Name"]), {})
                a[int(r["Dispatch_Id"])] = a.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
                break
    for (k, c), per in agg.items():
        v = list(per.values())
        res[k][c] = {"mean": sum(v) / len(v), "max": max(v), "min": min(v), "dispatches": len(v)}
        # the sampler's team kernel is dispatched three times by the probe, in this order: surface-like clouds (32),
        # dense regime (32 clouds), dense regime (4 clouds) -- one block per workload beside the mean over all three
        if k == "mds_dense_team_kernel" and len(v) % 3 == 0:
            ids = sorted(per)
            for i, tag in enumerate(("surface_b32", "dense_b32", "dense_b4")):
                w = [per[d] for d in ids[i::3]]
                res.setdefault(f"{k}#{tag}", {})[c] = {"mean": sum(w) / len(w), "max": max(w), "min": min(w), "dispatches": len(w)}
res = {k: v for k, v in res.items() if v}
import ctypes
lib = ctypes.CDLL("sparenet_amd/libsparenet_hip.so")
lib.sn_build_id.restype = ctypes.c_char_p
res["_build_id"] = lib.sn_build_id().decode()   # bench.py only quotes counters taken on the build it runs
res["_units"] = "per-dispatch means; FETCH_SIZE / WRITE_SIZE in KB; SQ_ACTIVE_* / SQ_WAIT_* / SQ_WAVE_CYCLES in quad-cycles; GRBM_GUI_ACTIVE summed over the 8 XCDs"
json.dump(res, open(sys.argv[1], "w"), indent=1)
for k, v in res.items():
    if isinstance(v, dict):
        print(k, {c: f"{x['mean']:.4g}" for c, x in v.items()})
PY
rm -rf gpurun_out/pmcA gpurun_out/pmcB gpurun_out/pmcC gpurun_out/pmcD
