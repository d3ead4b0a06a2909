"""Per-kernel summary of one rocprofv3 --kernel-trace run (any workload): calls, live launches, live average, share.

    python tools/make_trace_summary.py <trace dir> <out.txt> "<header line 1>" ["<header line 2>" ...]

live = duration > 50 % of the kernel's longest launch (speculative CG launches that found their solve converged return
at the guard after a few microseconds).
"""
import collections, csv, glob, sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    dur = collections.defaultdict(list)
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name'].replace('void ', '')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    total = sum(sum(v) for v in dur.values())
    rows = []
    for k, v in dur.items():
        live = [x for x in v if x > 0.5 * max(v)]
        rows.append((sum(v), k, len(v), len(live), sum(live) / len(live)))
    rows.sort(reverse=True)
    with open(out, 'w') as o:
        for h in sys.argv[3:]:
            o.write('# ' + h + '\n')
        o.write(f"{'kernel':70s} {'calls':>6s} {'n_live':>7s} {'live_avg_us':>12s} {'pct':>6s}\n")
        for tot, k, n, nl, la in rows:
            if tot / total < 0.001:
                continue
            o.write(f"{k[:70]:70s} {n:6d} {nl:7d} {la:12.1f} {100 * tot / total:6.1f}\n")
    print(open(out).read())


if __name__ == '__main__':
    main()
