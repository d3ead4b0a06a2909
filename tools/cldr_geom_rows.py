"""Rows of tools/cldr_geom_ab.sh: live launch time and HBM bytes ((2 * FETCH_SIZE + WRITE_SIZE) KiB, 16 B/lane loads) of the
streaming kernels of one bench run.   python tools/cldr_geom_rows.py <dir with stats/ fetch/ write/>"""
import collections, csv, glob, re, statistics as st, sys


def short(n):
    n = n.replace('void ', '')
    m = re.search(r'k_cldr<(\w+), (\d+), (\w+)<[^>]*>, CldrSrc(\w+)<[^>]*>, (\d+), (\d+), (\d+), (\d+)', n)
    if m:
        return f"k_cldr<{m.group(2)},{m.group(3)},{m.group(4)},NW{m.group(5)},{m.group(6)}/{m.group(7)}/{m.group(8)}>"
    m = re.search(r'(k_rows|k_tile)<(\w+), (\d+), (\w+)(?:<[^>]*>)?, (\d+)', n)
    if m:
        return f"{m.group(1)}<{m.group(3)},{m.group(4)}{',Fold' if 'TileSrcFold' in n else ''}>"
    return n.split('(')[0][:40]


def live(v):
    m = max(v)
    lv = [x for x in v if x > 0.5 * m] if m > 0 else v
    return st.mean(lv), len(lv)


def main():
    d = sys.argv[1]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(d + '/stats/**/*kernel_trace.csv', recursive=True)[0])):
        dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    cnt = {}
    for name in ('fetch', 'write'):
        c = collections.defaultdict(list)
        for r in csv.DictReader(open(glob.glob(d + f'/{name}/**/*counter_collection.csv', recursive=True)[0])):
            c[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
        cnt[name] = c
    print(f"{'kernel':44s} {'n_live':>6s} {'live_us':>9s} {'total_ms':>9s} {'read_MB':>9s} {'write_MB':>9s} {'TB/s':>6s}")
    for k in sorted(dur, key=lambda k: -sum(dur[k]))[:8]:
        lm, ln = live(dur[k])
        f = live(cnt['fetch'][k])[0] * 2 * 1024 / 1e6 if k in cnt['fetch'] else 0.0
        w = live(cnt['write'][k])[0] * 1024 / 1e6 if k in cnt['write'] else 0.0
        print(f"{k:44s} {ln:6d} {lm:9.1f} {sum(dur[k]) / 1e3:9.2f} {f:9.1f} {w:9.1f} {(f + w) / lm:6.2f}")


main()
