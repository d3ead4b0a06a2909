"""Timeline of one rocprofv3 --kernel-trace run: for every kernel launch between two consecutive launches of a named anchor
kernel (default k_admm_lds) its start relative to the START of the anchor launch and its duration, averaged over all such
periods (the metric kernels of the LDS path run on a side stream BESIDE the next anchor launch: offsets inside the anchor's
duration are overlap, not idle time).

    python tools/trace_timeline.py <trace dir> [anchor substring]

Used to see what the period of a cfg2 launch (7 ADMM iterations of the whole batch) consists of beside the k_admm_lds launch itself.
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
            acc[j - a][0] += (rows[j][0] - rows[a][0]) / 1e3
            acc[j - a][1] += (rows[j][1] - rows[j][0]) / 1e3
    print(f"# {n} periods of {len(seq)} launches, anchor '{anchor}', mean period {period / n:.1f} us")
    print(f"{'kernel':50s} {'start_us':>14s} {'duration_us':>12s}")
    td = 0.0
    for name, (g, du) in zip(seq, acc):
        print(f"{name:50s} {g / n:14.1f} {du / n:12.1f}")
        td += du / n
    print(f"{'sum of durations':50s} {'':14s} {td:12.1f}")


if __name__ == '__main__':
    main()
