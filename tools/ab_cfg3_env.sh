#!/bin/bash
# A/B of one environment knob on the cfg3 workload (streaming path) on ONE box: tools/ab_cfg3_env.sh <rounds> <VAR> <v1> <v2> ...
rounds=$1; var=$2; shift 2
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    env $var=$v python bench.py --workload cfg3 --steps 6 --warmup 1 --no-cpu-baseline > /tmp/ab_bench.log 2>&1
    python - "$var=$v" <<'P'
import json, sys
d = json.loads(open("/tmp/ab_bench.log").read().strip().splitlines()[-1])
r = d['roofline']
print(f"{sys.argv[1]:34s} {d['value']:10.1f} {d['ms_per_step']:8.3f}  spmm {r['avg_launch_us']:7.1f} us frac {r['frac']:.3f}  cgupdate {r.get('cg_update_kernel_GBs', 0):7.1f} GB/s", flush=True)
P
  done
done
