#!/bin/bash
# GPU call 1 of round 3: parity of the two-barrier auction, A/B against round 2's kernel, phases, bench, strong share
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
echo "== quick parity + timing (new)"; timeout 900 python tools/emd_ab.py --parity --parity32 > $O/emd_new.txt 2>&1; tail -25 $O/emd_new.txt
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/tests.log; tail -8 $O/tests.log
echo "== A/B per batch size"
for spec in "new::" "old:tools/ab/lib_old.so:" "new_legacy_geom::SN_EMD_GEOM=1" "new_G16::SN_EMD_G=16" "new_G8::SN_EMD_G=8" "new_safe::SN_EMD_SAFE=1"; do
  name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; envs=${rest#*:}
  echo "-- $name"
  env $envs AB_LIB=$lib AB_BS=32,16,8,4,2,1 timeout 600 python tools/emd_ab.py 2>&1 | grep "ms per call" | tee $O/ab_$name.txt
done
echo "== phases"
SN_EMD_DIAG=2 AB_BS=32 timeout 600 python tools/emd_ab.py > $O/phases_b32.txt 2>&1; tail -16 $O/phases_b32.txt
SN_EMD_DIAG=2 AB_BS=4 AB_DIAG_B=4 timeout 600 python tools/emd_ab.py > $O/phases_b4.txt 2>&1; tail -16 $O/phases_b4.txt
SN_EMD_DIAG=2 AB_BS=4 AB_DIAG_B=4 AB_LIB=tools/ab/lib_old.so timeout 600 python tools/emd_ab.py > $O/phases_b4_old.txt 2>&1; tail -16 $O/phases_b4_old.txt
echo "== bench"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','depthmaps_per_sec','segments_ms_rank0','sequential_ms_per_step_rank0','depthmaps_per_sec_literal_radii')}); print(d['roofline'])"
echo "== strong share"
timeout 600 python tools/strong_share.py 2>&1 | tee $O/strong_share.txt
echo "== traffic calibration"
timeout 600 tools/traffic_calibration.sh $O/traffic_calibration.json
