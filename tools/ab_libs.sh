#!/bin/bash
# A/B of library builds on ONE box: tools/ab_libs.sh <rounds> <lib A> <lib B> [...]  -- cfg2 bench, alternating, prints
# sample-iterations/s, ms per step and the average k_admm_lds launch of every run (different boxes differ by 1-2 %)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    MGADMM_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-cfg3-leg --warmup 5 > /tmp/ab_bench.log 2>&1
    python - "$lib" <<'P'
import json, sys
d = json.loads(open("/tmp/ab_bench.log").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:40s} {d['value']:10.0f} {d['ms_per_step']:8.4f} {d['roofline']['avg_iteration_us']:8.1f}", flush=True)
P
  done
done
