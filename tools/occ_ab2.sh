#!/bin/bash
# usage (GPU box): tools/occ_ab2.sh <lib> -- strong shares and the delayed renderer with the auction first
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'emd live us', round(d['kernels_rank0']['emd_auction']['avg_us']), 'gather us', round(d['kernels_rank0']['p2i_max_splat']['avg_us']))"; }
v=$1
echo "== strong share, default order, default lib"; python tools/strong_share.py 2>&1 | cut -c1-80
echo "== strong share, auction first, lib $v"; AB_LIB=$v BENCH_ORDER=auction_first python tools/strong_share.py 2>&1 | cut -c1-80
echo "== strong share, auction first, default lib"; BENCH_ORDER=auction_first python tools/strong_share.py 2>&1 | cut -c1-80
for d in 100000 300000 1000000; do
  echo -n "lib $v auction_first delay=$d: "
  AB_LIB=$v BENCH_ORDER=auction_first BENCH_RENDER_DELAY=$d timeout 300 python bench.py --no-cpu-baseline --no-other-ops --no-network-steps --no-literal-radii --steps 30 --warmup 5 2>/dev/null | line
done
