#!/bin/bash
# round 6, first GPU call: state of the tree + where p2i_max_bwd_accum and the dense sampler's round spend their time
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd); O=gpurun_out/r06_a; mkdir -p $O; export TMPDIR=/tmp
echo "== gpu tests"; timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $O/gpu_tests.txt
echo "== render kernels: base and the accumulate experiments (1: no global flush, 2: + no LDS adds, 3: + no table, 4: + no gathers)"
for v in base accx1 accx2 accx3 accx4; do
  if [ $v = base ]; then unset AB_LIB; else export AB_LIB=tools/ab/lib_$v.so; fi
  echo "-- $v"; (python tools/render_probe.py; KTOP=9 tools/kstats.sh tools/render_probe.py) 2>&1 | grep -v "amdgpu\|^E2026\|^W2026" | grep "render fwd\|p2i_\|depth_\|calls"
done 2>&1 | tee $O/render_accum_experiments.txt
unset AB_LIB
echo "== dense sampler stamps"; AB_LIB=tools/ab/lib_mdsstamps.so timeout 600 python tools/mds_dense_stamps.py 2>&1 | grep "mds dense" | tee $O/mds_dense_stamps.txt
echo "== gather stats"; AB_LIB=tools/ab/lib_diag.so timeout 300 python tools/gather_stats.py 2>&1 | grep radii | tee $O/gather_stats.txt
echo "== PMC"; timeout 1200 tools/pmc_all.sh $O/pmc_all_kernels.json > $O/pmc.log 2>&1; tail -16 $O/pmc.log | cut -c1-600
echo done > $O/done.txt
