#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
echo "== host overhead"; python tools/host_overhead.py 2>&1 | grep -v amdgpu; HO_B=4 python tools/host_overhead.py 2>&1 | grep -v amdgpu
echo "== launcher: 2 ranks on one GPU over gloo"
BENCH_DEBUG_SHARED_GPU=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --no-roofline > $O/bench_gpus2_shared.json 2> $O/bench_gpus2.err; tail -3 $O/bench_gpus2.err | cut -c1-300; python -c "
import json; d=json.load(open('$O/bench_gpus2_shared.json')); print({k: d[k] for k in ('n_gpus','rccl_ranks','scaling','value','ms_per_step','other_scaling')}); print(d['config'])"
echo "== refuses fewer GPUs than asked"; python bench.py --gpus 2 2>&1 | tail -1
echo "== bench (network steps)"; timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err | cut -c1-300; python -c "
import json; d=json.load(open('$O/bench.json')); print(d.get('network_steps_rank0')); print(d['ms_per_step'])"
