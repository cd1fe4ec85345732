#!/bin/bash
# Collects everything profiles/r06_<tag>_* quotes, on the GPU box, in one call:  tools/r6_profiles.sh <tag>
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
TAG=${1:-e}
O=gpurun_out/r06_$TAG; mkdir -p $O
export TMPDIR=/tmp
echo "== gpu tests"; timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/gpu_tests.txt
echo "== PMC (four counter-only passes over tools/step_probe.py)"
timeout 900 tools/pmc_all.sh profiles/r06_${TAG}_pmc_all_kernels.json > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-400
cp profiles/r06_${TAG}_pmc_all_kernels.json $O/pmc_all_kernels.json
echo "== bench (quotes the counters just taken: same build)"
timeout 1800 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err | cut -c1-300
python - <<PY
import json
d = json.load(open("$O/bench.json"))
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step", "depthmaps_per_sec", "depthmaps_per_sec_literal_radii", "sequential_ms_per_step_rank0")})
print("config", d["config"].get("order"), d["config"].get("streams"), d["config"].get("schedule_table_ms"))
print("segments", d["segments_ms_rank0"])
print("roofline", {k: r.get(k) for k in ("kernel", "bound", "achieved", "frac", "frac_basis", "algorithmic_frac", "wait_frac", "valu_busy", "mfma_busy", "traffic", "avg_launch_us", "counters_from")}, "live", r.get("live"))
for k in ("chamfer_fwd", "p2i_gather_max"):
    print(k, {a: r[k].get(a) for a in ("frac", "algorithmic_frac", "valu_busy", "wait_frac", "avg_launch_us", "search_kernel_avg_us", "traffic") if a in r[k]})
print("mds_team", {t: {a: d["roofline"]["mds_team"][t].get(a) for a in ("ms", "frac", "algorithmic_frac", "valu_busy", "wait_frac", "counters_from")} for t in ("surface", "dense")} if "mds_team" in d["roofline"] else None)
print("literal", d.get("literal_radii"))
print("emd regimes", {k: round(v, 3) for k, v in d.get("emd_regimes_rank0", {}).items() if k != "note"})
ns = d.get("network_steps_rank0", {})
print("network steps", {k: round(v, 1) for k, v in ns.items() if k.startswith("step_ms")}, {k: (round(v["min"], 1), round(v["max"], 1)) for k, v in ns.get("spread_ms", {}).items()})
print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind", "combined_speedup_vs_all_cores", "combined_speedup_vs_one_thread")})
PY
echo "== rocprofv3 --kernel-trace --stats of the bench command"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst -- python $R/bench.py --no-other-ops --no-cpu-baseline --no-network-steps --no-literal-radii --no-emd-regimes --steps 10 --warmup 2 > /dev/null 2>&1; cd $R
f=$(find gpurun_out/kst -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; head -8 $f | cut -c1-160; rm -rf gpurun_out/kst
echo "== emd regimes: this build, round 5 library if tools/ab/lib_r5.so travels along, phases"
{ echo "# this build"; timeout 600 python tools/emd_regimes.py 2>&1 | grep regime
  if [ -f tools/ab/lib_r5.so ]; then echo "# round 5 library (tools/ab/lib_r5.so)"; AB_LIB=tools/ab/lib_r5.so timeout 600 python tools/emd_regimes.py 2>&1 | grep regime; fi; } | tee $O/emd_regimes.txt
for bb in 32 4; do SN_EMD_DIAG=2 AB_BS=$bb timeout 600 python tools/emd_regimes.py uniform scatter untrained 2>&1 | grep -v amdgpu > $O/emd_phases_b$bb.txt; done; grep "regime\|sum over iterations of mean" $O/emd_phases_b32.txt | cut -c1-200
{ echo "uniform cubes:"; AB_BS=32,16,8,4,2,1 timeout 600 python tools/emd_ab.py 2>&1 | grep "ms per call"; } | tee $O/emd_per_batch.txt
echo "== strong share"; timeout 600 python tools/strong_share.py 2>&1 | grep -v amdgpu | tee $O/strong_share.txt
echo "== host overhead"; (HO_B=4 timeout 300 python tools/host_overhead.py; HO_B=32 timeout 300 python tools/host_overhead.py) 2>&1 | grep -v amdgpu | tee $O/host_overhead.txt
echo "== render kernels"; (python tools/render_probe.py; KTOP=12 tools/kstats.sh tools/render_probe.py) 2>&1 | grep -v "amdgpu\|^E2026" | tee $O/render_kernels.txt | tail -12
echo "== dense sampler stamps"; [ -f tools/ab/lib_mdsstamps.so ] && AB_LIB=tools/ab/lib_mdsstamps.so timeout 600 python tools/mds_dense_stamps.py 2>&1 | grep "mds dense" | tee $O/mds_dense_stamps.txt
echo "== sampler on surface clouds"; timeout 600 python tools/mds_surface.py --parity 2>&1 | grep "mds surface" | tee $O/mds_surface.txt
echo "== sampler: dense / intermediate / surface regimes per batch (teams from cut^2 > 0.075 diag^2 on)"; timeout 600 python tools/mds_ab.py 2>&1 | grep "^mds" | tee $O/mds_regimes.txt
echo "== network steps: steady-state kernel tables"
cd /tmp
for cfg in config4 config5; do
  for st in random_init trained_stand_in_damped; do
    rm -rf /tmp/prof_$cfg
    NS_WARMUP=3 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$cfg -o $cfg -- python $R/tools/net_step.py $cfg $st 6 2>&1 | grep "ms per step" > $R/$O/network_${cfg}_${st}_steady.txt
    f=$(find /tmp/prof_$cfg -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python $R/tools/steady_stats.py "$f" xor 6 16 >> $R/$O/network_${cfg}_${st}_steady.txt 2>&1
    head -8 $R/$O/network_${cfg}_${st}_steady.txt | cut -c1-200
  done
done
cd $R
echo "== launcher"; BENCH_DEBUG_SHARED_GPU=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-roofline 2>/dev/null > $O/bench_gpus2.out; python -c "
import json; t=open('$O/bench_gpus2.out').read(); t=t[t.index('{\"metric\"'):]; d=json.loads(t[:t.rindex('}')+1]); json.dump(d, open('$O/bench_gpus2_shared_gpu.json','w')); print({k: d[k] for k in ('n_gpus','scaling','ms_per_step')}, d['rccl_ranks']['backend'], d['other_scaling']['scaling'], d['other_scaling']['ms_per_step'])"
echo done > $O/done.txt
