"""Steady-state kernel statistics from a rocprofv3 --kernel-trace CSV: everything AFTER the marker kernel (the script
under the profiler launches one distinctive kernel when its warm-up is over -- MIOpen's find phase, JIT compiles and the
first allocations then stay out of the table).  Prints per kernel: ms per step, calls per step, share; then the busy /
idle split of the measured window.

    python tools/steady_stats.py <kernel_trace.csv> <marker substring> <steps> [top]"""
import csv, sys

path, marker, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 25
rows = list(csv.DictReader(open(path)))
name_k = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
t0 = max((int(r["End_Timestamp"]) for r in rows if marker.lower() in r[name_k].lower()), default=None)
if t0 is None:
    raise SystemExit(f"marker kernel '{marker}' not found among {len(rows)} dispatches")
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
agg = {}
for r in sel:
    a = agg.setdefault(r[name_k], [0, 0])
    a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a[1] += 1
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    busy += cur_e - cur_s
window = iv[-1][1] - t0 if iv else 0
total = sum(v[0] for v in agg.values())
print(f"steady state: {len(sel)} dispatches in {window / 1e6:.1f} ms = {steps} steps of {window / 1e6 / steps:.1f} ms; "
      f"GPU busy {busy / 1e6 / steps:.1f} ms per step ({100.0 * busy / max(window, 1):.0f} %), idle {((window - busy) / 1e6 / steps):.1f} ms; "
      f"sum of kernel durations {total / 1e6 / steps:.1f} ms per step (overlapping streams count twice)")
print(f"{'ms/step':>9} {'calls/step':>10} {'%':>6}  kernel")
for name, (ns, calls) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{ns / 1e6 / steps:9.2f} {calls / steps:10.1f} {100.0 * ns / max(total, 1):6.1f}  {name[:150]}")
rest = sorted(agg.items(), key=lambda kv: -kv[1][0])[top:]
print(f"{sum(v[0] for _, v in rest) / 1e6 / steps:9.2f} {sum(v[1] for _, v in rest) / steps:10.1f} {100.0 * sum(v[0] for _, v in rest) / max(total, 1):6.1f}  ({len(rest)} other kernels)")
