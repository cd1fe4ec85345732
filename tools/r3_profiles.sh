#!/bin/bash
# Collects everything profiles/r03_<tag>_* quotes, on the GPU box, in one call:  tools/r3_profiles.sh <tag>
cd ${GRAFT_REPO_ROOT:-.}
TAG=${1:-a}
O=gpurun_out/r03_$TAG; mkdir -p $O
export TMPDIR=/tmp
echo "== gpu tests"; timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/gpu_tests.txt
echo "== PMC (four counter-only passes over tools/step_probe.py)"
timeout 900 tools/pmc_all.sh profiles/r03_${TAG}_pmc_all_kernels.json > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-400
cp profiles/r03_${TAG}_pmc_all_kernels.json $O/pmc_all_kernels.json
echo "== bench (quotes the counters just taken: same build)"
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err | cut -c1-300
python - <<PY
import json
d = json.load(open("$O/bench.json"))
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step", "depthmaps_per_sec", "depthmaps_per_sec_literal_radii", "sequential_ms_per_step_rank0")})
print("segments", d["segments_ms_rank0"])
print("roofline", {k: r.get(k) for k in ("kernel", "achieved", "frac", "algorithmic_frac", "wait_frac", "valu_busy", "mfma_busy", "traffic", "avg_launch_us", "counters_from")})
for k in ("chamfer_fwd", "p2i_gather_max", "mds_clustered"):
    print(k, {a: r[k].get(a) for a in ("frac", "algorithmic_frac", "valu_busy", "wait_frac", "avg_launch_us", "traffic") if a in r[k]})
print("network steps", d.get("network_steps_rank0"))
print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind", "combined_speedup_vs_all_cores", "combined_speedup_vs_one_thread")})
PY
echo "== rocprofv3 --kernel-trace --stats of the bench command"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/kst -- python $OLDPWD/bench.py --no-other-ops --no-cpu-baseline --no-network-steps --no-literal-radii --steps 10 --warmup 2 > /dev/null 2>&1; cd $OLDPWD
f=$(find gpurun_out/kst -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; head -8 $f | cut -c1-160; rm -rf gpurun_out/kst
echo "== emd phases + per batch size"
SN_EMD_DIAG=2 AB_BS=32 AB_DIAG_B=32 timeout 600 python tools/emd_ab.py 2>&1 | grep -v amdgpu > $O/emd_phases_b32.txt; tail -2 $O/emd_phases_b32.txt | cut -c1-250
SN_EMD_DIAG=2 AB_BS=4 AB_DIAG_B=4 timeout 600 python tools/emd_ab.py 2>&1 | grep -v amdgpu > $O/emd_phases_b4.txt; tail -2 $O/emd_phases_b4.txt | cut -c1-250
AB_BS=32,16,8,4,2,1 timeout 600 python tools/emd_ab.py 2>&1 | grep "ms per call" | tee $O/emd_per_batch.txt
SN_EMD_SCAN=0 AB_BS=32,16,8,4,2,1 timeout 600 python tools/emd_ab.py 2>&1 | grep "ms per call" | sed 's/^/SN_EMD_SCAN=0 (group search in every iteration): /' | tee -a $O/emd_per_batch.txt
SN_EMD_SCAN=0 SN_EMD_DIAG=2 AB_BS=32 AB_DIAG_B=32 timeout 600 python tools/emd_ab.py 2>&1 | grep -v amdgpu > $O/emd_phases_b32_group_search_only.txt; tail -1 $O/emd_phases_b32_group_search_only.txt | cut -c1-250
echo "== strong share"; timeout 600 python tools/strong_share.py 2>&1 | grep -v amdgpu | tee $O/strong_share.txt
echo "== mds teams"; timeout 600 python tools/mds_ab.py 2>&1 | grep -v amdgpu | tee $O/mds_teams.txt; SN_MDS_G=1 timeout 600 python tools/mds_ab.py 2>&1 | grep -v amdgpu | sed 's/^/teams off: /' | tee -a $O/mds_teams.txt
echo "== render kernels"; (python tools/render_probe.py; KTOP=12 tools/kstats.sh tools/render_probe.py) 2>&1 | grep -v "amdgpu\|^E2026" | tee $O/render_kernels.txt | tail -12
echo "== chamfer collapsed"; python tools/chamfer_collapsed.py 2>&1 | grep chamfer | tee $O/chamfer_collapsed.txt
echo "== traffic calibration"; timeout 600 tools/traffic_calibration.sh $O/traffic_calibration.json | tee $O/traffic_calibration.txt
echo "== launcher"; BENCH_DEBUG_SHARED_GPU=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-roofline 2>/dev/null | grep '^{' > $O/bench_gpus2_shared_gpu.json; python -c "
import json; d=json.load(open('$O/bench_gpus2_shared_gpu.json')); print({k: d[k] for k in ('n_gpus','scaling','ms_per_step')}, d['rccl_ranks']['backend'], d['other_scaling']['scaling'], d['other_scaling']['ms_per_step'])"
