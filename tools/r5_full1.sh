#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5_full1; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"])
print("emd_regimes", {k: round(v, 3) for k, v in d.get("emd_regimes_rank0", {}).items() if k != "note"})
ns = d.get("network_steps_rank0", {})
print("network", {k: round(v, 1) for k, v in ns.items() if k.startswith("step_ms")})
print("spread", {k: (round(v["min"],1), round(v["max"],1)) for k, v in ns.get("spread_ms", {}).items()})
r = d["roofline"]; print("roofline frac", r.get("frac"), r.get("frac_basis"), r.get("avg_launch_us"), "live", r.get("live"))
print("chamfer", {k: r["chamfer_fwd"].get(k) for k in ("avg_launch_us", "search_kernel_avg_us", "frac")})
print("other", d.get("other_ops_ms_rank0"))
PY
