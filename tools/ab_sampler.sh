for r in 1 2 3; do
  for f in "" "--device-state"; do
    python bench.py --no-cpu-baseline --no-cfg3-leg --warmup 5 $f > /tmp/ab_bench.log 2>&1
    python - "flag=$f" <<'P'
import json, sys
d = json.loads(open("/tmp/ab_bench.log").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:30s} {d['value']:10.0f} {d['ms_per_step']:8.4f} {d['roofline']['avg_iteration_us']:8.1f} {d.get('device_state') and d['device_state'].get('power_w')}", flush=True)
P
  done
done
