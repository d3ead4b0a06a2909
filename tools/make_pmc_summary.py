"""Per-kernel SQ counter shares from rocprofv3 --pmc passes (8 SQ counters per pass; pass directories as arguments).

    python tools/make_pmc_summary.py profiles/rNN/NAME.txt <dir_pass1> [<dir_pass2> ...]

Counters are summed over waves in quad-cycles; what matters are the ratios to SQ_WAVE_CYCLES (per-wave shares) and the
bank-conflict share of the LDS-active cycles.  Mean over live dispatches (counter > 50 % of the kernel's maximum).
"""
import collections, csv, glob, re, statistics as st, sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])


def short(n):
    n = n.replace('void ', '')
    m = re.search(r'k_cldr<(\w+), (\d+), (\w+)<[^>]*>, CldrSrc(\w+)', n)
    if m:
        return f"k_cldr<{m.group(3)},{m.group(4)}>"
    m = re.search(r'(k_rows|k_tile)<(\w+), (\d+), (\w+)(?:<[^>]*>)?, (\d+)', n)
    if m:
        return f"{m.group(1)}<{m.group(4)},GW{m.group(5)}{',Fold' if 'TileSrcFold' in n else ''}>"
    m = re.search(r'k_admm_lds<(\d+), (\w+), (\d+), (\w+), (\d+), (\d+), (\w+), (-?\d+)>', n)
    if m:
        return f"k_admm_lds<TPG{m.group(1)},{'uniform' if m.group(5) != '0' else 'ragged'}>"
    return n.split('(')[0][:40]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    val = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
        for r in csv.DictReader(open(f)):
            val[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    def lm(v):
        m = max(v)
        lv = [x for x in v if x > 0.5 * m] if m > 0 else v
        return st.mean(lv) if lv else 0.0
    want = [k for k in val if k.startswith(('k_cldr', 'k_tile', 'k_rows', 'k_admm'))]
    want.sort(key=lambda k: -lm(val[k].get('SQ_WAVE_CYCLES', [0])) * len(val[k].get('SQ_WAVE_CYCLES', [0])))
    L = ["# rocprofv3 --pmc SQ passes (mean over live dispatches; *_CYCLES / WAIT_* / ACTIVE_* are quad-cycles summed over waves)",
         f"{'kernel':34s} {'valu_busy%':>10s} {'wait_any%':>10s} {'wait_inst%':>10s} {'active%':>8s} {'lds_idx%':>9s} {'lds_conflict%':>13s} {'VALU/wave':>10s} {'LDS/wave':>9s} {'SALU/wave':>10s}"]
    for k in want[:14]:
        c = {n: lm(v) for n, v in val[k].items()}
        wc = c.get('SQ_WAVE_CYCLES', 0) or 1
        waves = c.get('SQ_WAVES', 0) or 1
        pct = lambda n: 100.0 * c.get(n, 0) / wc
        conf = 100.0 * c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 0), 1)
        L.append(f"{k:34s} {pct('SQ_ACTIVE_INST_VALU'):10.1f} {pct('SQ_WAIT_ANY'):10.1f} {pct('SQ_WAIT_INST_ANY'):10.1f} {pct('SQ_ACTIVE_INST_ANY'):8.1f} "
                 f"{pct('SQ_LDS_IDX_ACTIVE'):9.1f} {conf:13.1f} {c.get('SQ_INSTS_VALU', 0) / waves:10.0f} {c.get('SQ_INSTS_LDS', 0) / waves:9.0f} {c.get('SQ_INSTS_SALU', 0) / waves:10.0f}")
    open(out, 'w').write("\n".join(L) + "\n")
    print("\n".join(L))


main()
