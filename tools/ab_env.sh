#!/bin/bash
# A/B of one environment knob on ONE box: tools/ab_env.sh <rounds> <VAR> <value A> <value B> [...]  (cfg2 bench, alternating)
rounds=$1; var=$2; shift 2
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    env $var=$v python bench.py --no-cpu-baseline --no-cfg3-leg --warmup 5 > /tmp/ab_bench.log 2>&1
    python - "$var=$v" <<'P'
import json, sys
d = json.loads(open("/tmp/ab_bench.log").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:34s} {d['value']:10.0f} {d['ms_per_step']:8.4f} {d['roofline']['avg_iteration_us']:8.1f}  cg {d['config']['mean_cg_iters_x_zu_zd']}", flush=True)
P
  done
done
