#!/bin/bash
# Collects everything profiles/r04_<tag>_* quotes, on the GPU box, in one call:  tools/r4_profiles.sh <tag>
cd ${GRAFT_REPO_ROOT:-.}
R=$(pwd)
TAG=${1:-a}
O=gpurun_out/r04_$TAG; mkdir -p $O
export TMPDIR=/tmp
echo "== gpu tests"; timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/gpu_tests.txt
echo "== PMC (four counter-only passes over tools/step_probe.py)"
timeout 900 tools/pmc_all.sh profiles/r04_${TAG}_pmc_all_kernels.json > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-400
cp profiles/r04_${TAG}_pmc_all_kernels.json $O/pmc_all_kernels.json
echo "== bench (quotes the counters just taken: same build)"
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err | cut -c1-300
python - <<PY
import json
d = json.load(open("$O/bench.json"))
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step", "depthmaps_per_sec", "depthmaps_per_sec_literal_radii", "sequential_ms_per_step_rank0")})
print("config", d["config"].get("order"), d["config"].get("streams"))
print("segments", d["segments_ms_rank0"])
print("roofline", {k: r.get(k) for k in ("kernel", "bound", "achieved", "frac", "algorithmic_frac", "wait_frac", "valu_busy", "mfma_busy", "traffic", "avg_launch_us", "bracket_avg_us", "queue_wait_avg_us", "counters_from")}, "isolated", r.get("isolated"))
for k in ("chamfer_fwd", "p2i_gather_max", "mds_clustered"):
    print(k, {a: r[k].get(a) for a in ("frac", "algorithmic_frac", "valu_busy", "wait_frac", "avg_launch_us", "traffic") if a in r[k]})
print("literal", d.get("literal_radii"))
print("network steps", d.get("network_steps_rank0"))
print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind", "combined_speedup_vs_all_cores", "combined_speedup_vs_one_thread")})
PY
echo "== rocprofv3 --kernel-trace --stats of the bench command"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst -- python $R/bench.py --no-other-ops --no-cpu-baseline --no-network-steps --no-literal-radii --steps 10 --warmup 2 > /dev/null 2>&1; cd $R
f=$(find gpurun_out/kst -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; head -8 $f | cut -c1-160; rm -rf gpurun_out/kst
echo "== stream order A/B"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_rank0']; ki=d['kernels_isolated_rank0']; print(round(d['ms_per_step'],3), 'ms per step; one stream', round(d['sequential_ms_per_step_rank0'],3), '; auction bracket / execution window / isolated us', round(k['emd_auction']['avg_us']), round(k['emd_auction_exec']['avg_us']), round(ki['emd_auction']['avg_us']), '; gather live / isolated us', round(k['p2i_max_splat']['avg_us']), round(ki['p2i_max_splat']['avg_us']))"; }
BA="--no-cpu-baseline --no-other-ops --no-network-steps --no-literal-radii --steps 30 --warmup 8"
for o in chain auction_first; do echo -n "BENCH_ORDER=$o: "; BENCH_ORDER=$o timeout 300 python bench.py $BA 2>/dev/null | line; done | tee $O/order_ab.txt
echo "== emd phases + per batch size"
SN_EMD_DIAG=2 AB_BS=32 AB_DIAG_B=32 timeout 600 python tools/emd_ab.py 2>&1 | grep -v amdgpu > $O/emd_phases_b32.txt; tail -2 $O/emd_phases_b32.txt | cut -c1-250
SN_EMD_DIAG=2 AB_BS=4 AB_DIAG_B=4 timeout 600 python tools/emd_ab.py 2>&1 | grep -v amdgpu > $O/emd_phases_b4.txt; tail -2 $O/emd_phases_b4.txt | cut -c1-250
{ echo "uniform cubes:"; AB_BS=32,16,8,4,2,1 timeout 600 python tools/emd_ab.py 2>&1 | grep "ms per call"
  echo "prediction = ground truth on a sphere + 1 % noise (AB_DATA=surface):"; AB_DATA=surface AB_BS=32,4 timeout 600 python tools/emd_ab.py 2>&1 | grep "ms per call"
  echo "prediction scattered +-0.3 around a sphere (AB_DATA=scatter):"; AB_DATA=scatter AB_BS=32,4 timeout 600 python tools/emd_ab.py 2>&1 | grep "ms per call"; } | tee $O/emd_per_batch.txt
echo "== strong share"; timeout 600 python tools/strong_share.py 2>&1 | grep -v amdgpu | tee $O/strong_share.txt
echo "== host overhead"; (HO_B=4 timeout 300 python tools/host_overhead.py; HO_B=32 timeout 300 python tools/host_overhead.py) 2>&1 | grep -v amdgpu | tee $O/host_overhead.txt
echo "== render kernels"; (python tools/render_probe.py; KTOP=12 tools/kstats.sh tools/render_probe.py) 2>&1 | grep -v "amdgpu\|^E2026" | tee $O/render_kernels.txt | tail -12
echo "== network steps: steady-state kernel tables + host profile"
cd /tmp
for cfg in config4 config5; do
  rm -rf /tmp/prof_$cfg
  NS_WARMUP=3 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$cfg -o $cfg -- python $R/tools/net_step.py $cfg trained_stand_in 6 2>&1 | grep "ms per step" > $R/$O/network_${cfg}_steady.txt
  f=$(find /tmp/prof_$cfg -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python $R/tools/steady_stats.py "$f" xor 6 24 >> $R/$O/network_${cfg}_steady.txt 2>&1
  head -12 $R/$O/network_${cfg}_steady.txt | cut -c1-200
done
cd $R
for cfg in config4 config5; do timeout 300 python tools/net_step.py $cfg random_init 7 2>&1 | grep "ms per step"; done | tee $O/network_random_init.txt
timeout 300 python tools/net_host_profile.py config5 trained_stand_in 2>&1 | grep -v amdgpu | head -40 > $O/config5_host_profile.txt; head -3 $O/config5_host_profile.txt
echo "== graph replay"
{ for a in "emd" "emd null+autofree" "chamfer"; do echo "-- raw HIP graph: $a"; SN_ALLOW_CAPTURE=1 timeout 60 tools/probe/graph_emd $a; done
  echo "-- torch.cuda.CUDAGraph (tools/capture_probe.py)"; timeout 600 python tools/capture_probe.py 2>&1 | grep -v amdgpu; } > $O/graph_replay.txt 2>&1; grep -c "equal to eager: 1" $O/graph_replay.txt
echo "== launcher"; BENCH_DEBUG_SHARED_GPU=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-roofline 2>/dev/null | grep '^{' > $O/bench_gpus2_shared_gpu.json; python -c "
import json; d=json.load(open('$O/bench_gpus2_shared_gpu.json')); print({k: d[k] for k in ('n_gpus','scaling','ms_per_step')}, d['rccl_ranks']['backend'], d['other_scaling']['scaling'], d['other_scaling']['ms_per_step'])"
echo done > $O/done.txt
