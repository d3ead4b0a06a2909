for cfg in "0 0" "1 0" "1 1" "0 0" "1 0" "1 1"; do set -- $cfg; MGADMM_LDS_ASYNC=$1 MGADMM_LDS_SIDE_PRIO=$2 python bench.py --no-cpu-baseline --no-cfg3-leg --warmup 5 > gpurun_out/s5_bench.log 2>&1; python - <<P
import json
d=json.loads(open("gpurun_out/s5_bench.log").read().strip().splitlines()[-1]); print("async=$1 hiprio=$2", round(d["value"]), round(d["ms_per_step"],4), round(d["roofline"]["avg_launch_us"],1))
P
done
