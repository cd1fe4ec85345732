#!/bin/bash
# round 4, first GPU call: the hygiene batch under the GPU tests, the graph-replay root-cause probes, and the
# co-residency / streaming-store A/B runs.  Everything goes to gpurun_out/r4c1/.
O=gpurun_out/r4c1; mkdir -p $O
export TMPDIR=/tmp
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_rank0']; ki=d['kernels_isolated_rank0']; print(round(d['ms_per_step'],3), 'seq', round(d['sequential_ms_per_step_rank0'],3), 'emd live/iso us', round(k['emd_auction']['avg_us']), round(ki['emd_auction']['avg_us']), 'gather live/iso us', round(k['p2i_max_splat']['avg_us']), round(ki['p2i_max_splat']['avg_us']), {a: round(v,2) for a,v in d['segments_ms_rank0'].items()})"; }
BA="--no-cpu-baseline --no-other-ops --no-network-steps --no-literal-radii --steps 30 --warmup 8"
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/gpu_tests.txt
{
  echo "== graph_lds dyn 140 KB"; timeout 60 tools/probe/graph_lds 140
  echo "== graph_lds dyn 48 KB"; timeout 60 tools/probe/graph_lds 48
  echo "== graph_lds static 144 KB"; timeout 60 tools/probe/graph_lds 140 static
} > $O/graph_lds.txt 2>&1
timeout 600 python tools/capture_probe.py > $O/capture_probe.txt 2>&1
{
  for lib in "" tools/ab/lib_nt.so; do
    echo "== lib ${lib:-default}: default order (two streams)"; AB_LIB=$lib timeout 300 python bench.py $BA 2>/dev/null | line
    echo "== lib ${lib:-default}: three streams"; AB_LIB=$lib BENCH_THREE_STREAMS=1 timeout 300 python bench.py $BA 2>/dev/null | line
    for occ in 4 5; do
      echo "== lib ${lib:-default}: auction_first, SN_EMD_OCC=$occ"; AB_LIB=$lib SN_EMD_OCC=$occ BENCH_ORDER=auction_first timeout 300 python bench.py $BA 2>/dev/null | line
    done
  done
} > $O/bench_ab.txt 2>&1
{
  for occ in 4 5; do echo "== emd per call, SN_EMD_OCC=$occ"; SN_EMD_OCC=$occ AB_BS=32,16,8,4 timeout 300 python tools/emd_ab.py 2>&1 | grep "per call"; done
} > $O/emd_occ.txt 2>&1
echo done > $O/done.txt
