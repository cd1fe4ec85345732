#!/bin/bash
# round 4, third GPU call: the gather's V2 path as the default + the backward's cheaper fixed-point conversion under the
# p2i tests; the auction as ONE 8-wave workgroup per CU (half of every SIMD's registers free) beside the renderer;
# kernel stats of config 4 / 5.
O=gpurun_out/r4c3; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_rank0']; ki=d['kernels_isolated_rank0']; print(round(d['ms_per_step'],3), 'seq', round(d['sequential_ms_per_step_rank0'],3), 'emd live/iso us', round(k['emd_auction']['avg_us']), round(ki['emd_auction']['avg_us']), 'gather live/iso us', round(k['p2i_max_splat']['avg_us']), round(ki['p2i_max_splat']['avg_us']), {a: round(v,2) for a,v in d['segments_ms_rank0'].items()})"; }
BA="--no-cpu-baseline --no-other-ops --no-network-steps --no-literal-radii --steps 30 --warmup 8"
( timeout 600 python -m pytest tests/test_p2i.py tests/test_fullsize.py tests/test_harness.py -m gpu -q 2>&1 | tail -5 ) > $O/gpu_tests_p2i.txt
{
  for lib in tools/ab/lib_w8.so tools/ab/lib_w8r.so; do
    echo "== $lib"; AB_LIB=$lib AB_BS=32,4 timeout 300 python tools/emd_ab.py --parity --parity32 2>&1 | grep "parity\|per call"
  done
} > $O/emd_w8.txt 2>&1
{
  echo "== default lib, default order"; timeout 300 python bench.py $BA 2>/dev/null | line
  echo "== default lib, auction_first"; BENCH_ORDER=auction_first timeout 300 python bench.py $BA 2>/dev/null | line
  for lib in tools/ab/lib_w8.so tools/ab/lib_w8r.so; do
    echo "== $lib, default order"; AB_LIB=$lib timeout 300 python bench.py $BA 2>/dev/null | line
    echo "== $lib, auction_first"; AB_LIB=$lib BENCH_ORDER=auction_first timeout 300 python bench.py $BA 2>/dev/null | line
  done
} > $O/bench_ab.txt 2>&1
( KTOP=16 bash tools/kstats.sh tools/render_probe.py ) > $O/render_kernels.txt 2>&1
cd /tmp
for cfg in config4 config5; do
  rm -rf /tmp/prof_$cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o $cfg -- python $R/tools/net_step.py $cfg trained_stand_in 5 2>&1 | grep "ms per step" > $R/$O/net_${cfg}.txt
  f=$(find /tmp/prof_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -60 "$f" > $R/$O/net_${cfg}_kernel_stats.csv
done
cd $R
echo done > $O/done.txt
