#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); O=gpurun_out/r06_b; mkdir -p $O; export TMPDIR=/tmp
echo "== p2i parity"; timeout 1200 python -m pytest tests/test_p2i.py tests/test_fullsize.py -m gpu -q -x -k "p2i or depth or render" 2>&1 | tail -3 | tee $O/p2i_tests.txt
echo "== render kernels per accumulate region"
for v in base reg2 reg4s1k reg8; do
  if [ $v = base ]; then unset AB_LIB; else export AB_LIB=tools/ab/lib_$v.so; fi
  echo "-- $v"; python tools/render_probe.py 2>&1 | grep "render fwd"; KTOP=6 tools/kstats.sh tools/render_probe.py 2>&1 | grep "p2i_\|calls"
done 2>&1 | tee $O/render_accum_regions.txt
unset AB_LIB
echo "== strong share with the measured schedule"; timeout 600 python tools/strong_share.py 2>&1 | grep -v amdgpu | tee $O/strong_share.txt
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline --no-network-steps --no-emd-regimes --no-literal-radii > $O/bench_quick.json 2> $O/bench.err; python - <<PY
import json
d = json.load(open("$O/bench_quick.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["config"]["order"], d["config"]["schedule_table_ms"], d["segments_ms_rank0"])
PY
echo done > $O/done.txt
