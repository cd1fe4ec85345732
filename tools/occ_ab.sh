#!/bin/bash
# usage (GPU box): tools/occ_ab.sh -- the auction with 128 / 96 / 80 VGPRs per wave (SN_EMD_OCC 4 / 5 / 6), alone and
# inside the step with the renderer beside it (BENCH_ORDER=auction_first)
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'seq', round(d['sequential_ms_per_step_rank0'],3), 'emd live us', round(d['kernels_rank0']['emd_auction']['avg_us']), 'gather us', round(d['kernels_rank0']['p2i_max_splat']['avg_us']), {k: round(v,2) for k,v in d['segments_ms_rank0'].items()})"; }
for v in "" "$@"; do
  echo "== lib ${v:-default}"
  AB_LIB=$v AB_BS=32,4 python tools/emd_ab.py 2>&1 | grep "per call"
  for order in "" auction_first; do
    echo -n "order=${order:-default}: "
    AB_LIB=$v BENCH_ORDER=$order timeout 300 python bench.py --no-cpu-baseline --no-other-ops --no-network-steps --no-literal-radii --steps 30 --warmup 5 2>/dev/null | line
  done
done
