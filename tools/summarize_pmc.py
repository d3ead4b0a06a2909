"""Summarise rocprofv3 output dirs: per-kernel mean duration (kernel_trace) and FETCH/WRITE KB (counter_collection)."""
import csv, collections, glob, re, statistics as st, sys
def short(n):
    m = re.search(r'k_rows<(\w+), (\d+), (\w+)<[^>]*>, (\d+)', n)
    if m: return f"k_rows<{m.group(1)},{m.group(2)},{m.group(3)},GW{m.group(4)}>"
    m = re.search(r'k_admm_lds<(\d+), (\w+)>', n)
    if m: return f"k_admm_lds<{m.group(1)},{m.group(2)}>"
    return n.split('(')[0].replace('void ', '')[:50]
d = sys.argv[1]
dur = collections.defaultdict(list)
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    break
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    if 'k_rows' not in k and 'k_admm' not in k: continue
    v = dur[k]; live = [x for x in v if x > 0.5 * max(v)]
    line = f"{k:40s} n={len(v):4d} live={len(live):4d} avg_us={st.mean(live):9.1f}"
    for c, vals in cnt.get(k, {}).items():
        lv = [x for x in vals if x > 0.5 * max(vals)] if max(vals) > 0 else [0]
        line += f"  {c}={st.mean(lv):.0f}"
    print(line)
