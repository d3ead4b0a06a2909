for us in 48 0 24 72 48 32; do MGADMM_LDS_STAGGER_US=$us python bench.py --no-cpu-baseline --no-cfg3-leg --warmup 5 > gpurun_out/s8_bench.log 2>&1; python - <<P
import json
d=json.loads(open("gpurun_out/s8_bench.log").read().strip().splitlines()[-1]); print("stagger_us=$us", round(d["value"]), round(d["ms_per_step"],4), round(d["roofline"]["avg_launch_us"],1))
P
done
