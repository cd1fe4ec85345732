#!/bin/bash
# usage (GPU box): tools/bench_ab.sh [tools/ab/lib_x.so ...]  -- short bench with the default library and each A/B build
for v in "" "$@"; do
  echo "== lib ${v:-default}"
  AB_LIB=$v python bench.py --no-cpu-baseline --no-other-ops --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['sequential_ms_per_step_rank0'],3), {k: round(v,2) for k,v in d['segments_ms_rank0'].items()})"
done
