"""Timeline of one rocprofv3 --kernel-trace run: for every kernel launch its start relative to the previous launch's end
(idle gap of the device between two kernels) and its duration, printed for the launches between two consecutive launches
of a named anchor kernel (default k_admm_lds), averaged over all such periods.

    python tools/trace_timeline.py <trace dir> [anchor substring]

Used to see what the 2.73 ms of a cfg2 ADMM iteration consist of beside the 2.57 ms k_admm_lds launch.
"""
import collections, csv, glob, sys


def main():
    d = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "k_admm_lds"
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '')) for r in csv.DictReader(open(f))]
    rows.sort()
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(idx) < 3:
        print("fewer than 3 anchor launches"); return
    # periods between consecutive anchors with the same kernel sequence as the most common one
    seqs = collections.Counter(tuple(r[2][:48] for r in rows[a:b]) for a, b in zip(idx[:-1], idx[1:]))
    seq = seqs.most_common(1)[0][0]
    acc = [[0.0, 0.0] for _ in seq]
    n = 0
    period = 0.0
    for a, b in zip(idx[:-1], idx[1:]):
        if tuple(r[2][:48] for r in rows[a:b]) != seq:
            continue
        n += 1
        period += (rows[b][0] - rows[a][0]) / 1e3
        for j in range(a, b):
            prev_end = rows[j - 1][1] if j > 0 else rows[j][0]
            acc[j - a][0] += (rows[j][0] - prev_end) / 1e3
            acc[j - a][1] += (rows[j][1] - rows[j][0]) / 1e3
    print(f"# {n} periods of {len(seq)} launches, anchor '{anchor}', mean period {period / n:.1f} us")
    print(f"{'kernel':50s} {'gap_before_us':>14s} {'duration_us':>12s}")
    tg = td = 0.0
    for name, (g, du) in zip(seq, acc):
        print(f"{name:50s} {g / n:14.1f} {du / n:12.1f}")
        tg += g / n; td += du / n
    print(f"{'sum':50s} {tg:14.1f} {td:12.1f}")


if __name__ == '__main__':
    main()
