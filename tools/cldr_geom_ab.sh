#!/bin/bash
# k_cldr tile geometries side by side on the cfg3 workload (run from the repo root on the GPU box):
#   tools/cldr_geom_ab.sh <outdir> <geom> [<geom> ...]
# per geometry: a kernel-trace pass, a FETCH_SIZE pass and a WRITE_SIZE pass of `bench.py --workload cfg3 --steps 3`, then
# live launch time and HBM bytes of the k_cldr instances (MGADMM_CLDR_GEOM, engine.h).
set -u
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CMD="python3 bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu-baseline --no-prof"
for g in "$@"; do
  W=/tmp/cg_$g; rm -rf $W; mkdir -p $W
  export MGADMM_CLDR_GEOM=$g MGADMM_TILE_STATS=1
  rocprofv3 --kernel-trace --stats -d $W/stats -o p --output-format csv -- $CMD > $W/stats.log 2>&1 || { echo "geom $g stats failed"; tail -5 $W/stats.log; exit 1; }
  rocprofv3 --pmc FETCH_SIZE -d $W/fetch -o p --output-format csv -- $CMD > $W/fetch.log 2>&1 || { echo "geom $g fetch failed"; exit 1; }
  rocprofv3 --pmc WRITE_SIZE -d $W/write -o p --output-format csv -- $CMD > $W/write.log 2>&1 || { echo "geom $g write failed"; exit 1; }
  { echo "== MGADMM_CLDR_GEOM=$g"; grep -h "cldr tiles" $W/stats.log | sort -u; grep -h '"metric"' $W/stats.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('sample-it/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2))";
    python3 tools/cldr_geom_rows.py $W; } | tee $out/geom$g.txt
done
