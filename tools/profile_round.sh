#!/bin/bash
# All rocprofv3 passes behind profiles/rNN/ in one GPU session (run from the repo root on the GPU box):
#   tools/profile_round.sh <rNN>
# kernel trace + stats, FETCH_SIZE pass, WRITE_SIZE pass (separate runs, as MI355X_MICROARCH.md prescribes), two SQ counter
# passes, a MGADMM_FOLD=0 stats pass (SpMM-only rows) and a 5-launch cfg2 trace (timeline), each of `python3 bench.py --steps 7 --warmup 1 --no-cpu-baseline --no-cfg4-leg --no-prof` (7 steps = ONE k_admm_lds launch of 7 iterations) + the FETCH_SIZE calibration; summaries are written to
# gpurun_out/profiles_<rNN>/ (copy them into profiles/<rNN>/) and profiles/traffic.json is rewritten there too.
set -u
r=$1
out=$PWD/gpurun_out/profiles_$r
mkdir -p $out/$r
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
W=/tmp/prof_$r; rm -rf $W; mkdir -p $W
CMD="python3 bench.py --steps 7 --warmup 1 --no-cpu-baseline --no-cfg4-leg --no-prof"
run() { name=$1; shift; echo "== $name"; rocprofv3 "$@" -d $W/$name -o p --output-format csv -- ${ALT:-$CMD} > $W/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $W/$name.log; exit 1; }; }
run stats --kernel-trace --stats
MGADMM_FOLD=0 run fold0 --kernel-trace --stats
# cfg2 alone, 64 iterations = 4 launches of 16: the period between two k_admm_lds launches (side-stream metric kernels included)
ALT="python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-cfg3-leg --no-prof" run timeline --kernel-trace
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT
bash tools/fetch_calib.sh $out/$r/fetch_calibration.json > $W/calib.log 2>&1 || { echo 'calibration failed'; tail -5 $W/calib.log; }
python tools/make_profile_summary.py $W $out/$r/bench_rocprof_summary.txt "$CMD" $out/$r/fetch_calibration.json > /dev/null || exit 1
python tools/make_pmc_summary.py $out/$r/pmc_sq_counters.txt $W/sq1 $W/sq2 > /dev/null || exit 1
python tools/trace_timeline.py $W/timeline > $out/$r/cfg2_iteration_timeline.txt
echo "written: $(ls $out/$r) + $out/traffic.json"
