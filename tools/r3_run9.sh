#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
echo "== strong share"; timeout 600 python tools/strong_share.py 2>&1 | grep -v amdgpu | tee $O/strong_share.txt
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
echo "== bench"; timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','depthmaps_per_sec','segments_ms_rank0','sequential_ms_per_step_rank0','depthmaps_per_sec_literal_radii')}); print(d['other_ops_ms_rank0'])"
