#!/bin/bash
# usage (GPU box): tools/traffic_calibration.sh <out.json>
# FETCH_SIZE / WRITE_SIZE of the known-byte kernels of tools/probe/traffic_probe.hip, two counter-only passes.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/tcalA -- $R/tools/probe/traffic_probe > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/tcalB -- $R/tools/probe/traffic_probe > /dev/null 2>&1
cd $R
python - "$OUT" <<'PY'
import csv, glob, json, sys
payload = 64 << 20
res = {}
for d, cname in (("tcalA", "FETCH_SIZE"), ("tcalB", "WRITE_SIZE")):
    fs = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        res.setdefault("_missing", []).append(d)
        continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != cname:
            continue
        k = r["Kernel_Name"].split("(")[0]
        a = agg.setdefault(k, {})
        a[int(r["Dispatch_Id"])] = a.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    for k, per in agg.items():
        v = list(per.values())
        res.setdefault(k, {})[cname + "_bytes"] = sum(v) / len(v) * 1024.0     # counters are in KB
for k, v in res.items():
    if k.startswith("_"):
        continue
    pay = payload / 16 if "scattered" in k else payload
    v["payload_bytes"] = pay
    if "FETCH_SIZE_bytes" in v:
        v["fetch_per_payload_byte"] = v["FETCH_SIZE_bytes"] / pay
    if "WRITE_SIZE_bytes" in v:
        v["write_per_payload_byte"] = v["WRITE_SIZE_bytes"] / pay
res["_units"] = ("per-launch means; *_stream: a 64 MiB buffer touched once; *_repeat: a 1 MiB window 64 times; "
                 "*_scattered: one 4-byte access per 64-byte line; raw counters (no x2 correction)")
json.dump(res, open(sys.argv[1], "w"), indent=1)
for k, v in sorted(res.items()):
    if isinstance(v, dict):
        print(f"{k:28s} fetch/payload {v.get('fetch_per_payload_byte', float('nan')):8.3f}  write/payload {v.get('write_per_payload_byte', float('nan')):8.3f}")
PY
rm -rf gpurun_out/tcalA gpurun_out/tcalB
