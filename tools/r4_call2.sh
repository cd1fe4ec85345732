#!/bin/bash
# round 4, second GPU call: full GPU tests after the sampler's vote fix, the gather's V2 build (parity + time), the
# capture probes through the Python wrapper, the auction-first order at every batch share, config 4 / 5 kernel stats.
O=gpurun_out/r4c2; mkdir -p $O
export TMPDIR=/tmp
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_rank0']; ki=d['kernels_isolated_rank0']; print(round(d['ms_per_step'],3), 'seq', round(d['sequential_ms_per_step_rank0'],3), 'emd live/iso us', round(k['emd_auction']['avg_us']), round(ki['emd_auction']['avg_us']), 'gather live/iso us', round(k['p2i_max_splat']['avg_us']), round(ki['p2i_max_splat']['avg_us']), {a: round(v,2) for a,v in d['segments_ms_rank0'].items()})"; }
BA="--no-cpu-baseline --no-other-ops --no-network-steps --no-literal-radii --steps 30 --warmup 8"
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/gpu_tests.txt
( AB_LIB=tools/ab/lib_v2.so timeout 600 python -m pytest tests/test_p2i.py tests/test_fullsize.py tests/test_networks.py tests/test_harness.py -m gpu -q -k "p2i or render or depth or views or network or gan" 2>&1 | tail -8 ) > $O/gpu_tests_p2i_v2.txt
timeout 900 python tools/capture_probe.py > $O/capture_probe.txt 2>&1
{
  for lib in "" tools/ab/lib_v2.so; do
    echo "== lib ${lib:-default}: default order"; AB_LIB=$lib timeout 300 python bench.py $BA 2>/dev/null | line
    echo "== lib ${lib:-default}: auction_first"; AB_LIB=$lib BENCH_ORDER=auction_first timeout 300 python bench.py $BA 2>/dev/null | line
  done
} > $O/bench_ab.txt 2>&1
{
  echo "== strong shares, default order"; timeout 300 python tools/strong_share.py 2>&1 | grep "N ="
  echo "== strong shares, auction_first"; BENCH_ORDER=auction_first timeout 300 python tools/strong_share.py 2>&1 | grep "N ="
  echo "== strong shares, auction_first, V2 gather"; AB_LIB=tools/ab/lib_v2.so BENCH_ORDER=auction_first timeout 300 python tools/strong_share.py 2>&1 | grep "N ="
} > $O/strong_share.txt 2>&1
cd /tmp
for cfg in config4 config5; do
  rm -rf /tmp/prof_$cfg
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cfg -o $cfg -- python $GRAFT_REPO_ROOT/tools/net_step.py $cfg trained_stand_in 5 > $GRAFT_REPO_ROOT/$O/net_${cfg}.txt 2>&1
  f=$(find /tmp/prof_$cfg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -45 "$f" > $GRAFT_REPO_ROOT/$O/net_${cfg}_kernel_stats.csv
done
cd $GRAFT_REPO_ROOT
for cfg in config4 config5; do
  NS_OVERLAP=0 timeout 300 python tools/net_step.py $cfg trained_stand_in 7 >> $O/net_no_overlap.txt 2>&1
  timeout 300 python tools/net_step.py $cfg random_init 7 >> $O/net_random_init.txt 2>&1
done
echo done > $O/done.txt
