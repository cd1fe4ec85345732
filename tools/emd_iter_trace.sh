#!/bin/bash
# per-iteration kernel durations of one 50-iteration EMD call at C2 (run on the GPU box)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_emd -- python $R/tools/emd_probe.py > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_emd/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "emd_" in r["Kernel_Name"]]
# last EMD call = after the last emd_init_kernel
last = max(i for i, r in enumerate(rows) if "emd_init" in r["Kernel_Name"])
rows = rows[last:]
def short(n):
    for k in ("bid_kernel", "resolve", "compact", "getmax", "assign", "init", "calcdist", "seed", "sbbox", "sort_count", "sort_scatter"):
        if k in n: return k
    return n[:30]
it, cur, t_prev = [], {}, None
for r in rows:
    k = short(r["Kernel_Name"]); d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if k == "bid_kernel" and cur:
        it.append(cur); cur = {}
    cur[k] = d; cur.setdefault("t0", int(r["Start_Timestamp"])); cur["t1"] = int(r["End_Timestamp"])
it.append(cur)
keys = [k for k in ("bid_kernel", "resolve", "getmax", "assign", "compact") if any(k in c for c in it)]
print("iter  " + "  ".join(f"{k[-12:]:>12}" for k in keys) + "   span_us")
for i, c in enumerate(it):
    if "t0" not in c: continue
    if i < 12 or i % 4 == 0:
        nxt = it[i + 1]["t0"] if i + 1 < len(it) and "t0" in it[i + 1] else c["t1"]
        print(f"{i:4d}  " + "  ".join(f"{c.get(k, 0):12.1f}" for k in keys) + f"   {(nxt - c['t0']) / 1e3:8.1f}")
tot = {k: sum(c.get(k, 0) for c in it) for k in keys}
print("totals ms:", {k: round(v / 1e3, 3) for k, v in tot.items()}, "call span ms", (rows[-1] and (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6))
PY
rm -rf gpurun_out/prof_emd
