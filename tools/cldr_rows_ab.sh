#!/bin/bash
# rows-per-tile limit of the k_cldr tile builder (MGADMM_CLDR_ROWS) on the cfg3 workload: kernel-trace pass per value.
#   tools/cldr_rows_ab.sh <outdir> <limit> [<limit> ...]      (0 = the geometry's cap)
set -u
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CMD="python3 bench.py --workload cfg3 --steps 3 --warmup 1 --no-cpu-baseline --no-prof"
for r in "$@"; do
  W=/tmp/cr_$r; rm -rf $W; mkdir -p $W
  export MGADMM_CLDR_ROWS=$r MGADMM_TILE_STATS=1
  rocprofv3 --kernel-trace --stats -d $W/stats -o p --output-format csv -- $CMD > $W/stats.log 2>&1 || { echo "rows $r failed"; tail -5 $W/stats.log; exit 1; }
  { echo "== MGADMM_CLDR_ROWS=$r"; grep -h "cldr tiles" $W/stats.log | sort -u; grep -h '"metric"' $W/stats.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('sample-it/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2))";
    python3 - $W <<'P'
import collections, csv, glob, sys, statistics as st
dur = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob(sys.argv[1] + '/stats/**/*kernel_trace.csv', recursive=True)[0])):
    n = r['Kernel_Name']
    if 'k_cldr' in n:
        dur['k_cldr Fold' if 'Fold' in n else ('k_cldr CgInit' if 'CgInit' in n else 'k_cldr other')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in dur.items():
    lv = [x for x in v if x > 0.5 * max(v)]
    print(f"{k:16s} n_live {len(lv):4d} live_us {st.mean(lv):8.1f}")
P
  } | tee $out/rows$r.txt
done
