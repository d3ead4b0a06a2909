"""Turn the three rocprofv3 passes of one bench command into the committed summary + profiles/traffic.json.

    python tools/make_profile_summary.py <dir> profiles/rNN/NAME.txt "<bench command>" [profiles/rNN/fetch_calibration.json]

expects  <dir>/stats  (--kernel-trace --stats), <dir>/fetch (--pmc FETCH_SIZE), <dir>/write (--pmc WRITE_SIZE):
separate runs, as MI355X_MICROARCH.md prescribes.  FETCH_SIZE correction PER KERNEL, by the width of its global loads
(round 3): the guide's "x 2" holds for 16 B/lane streaming reads; other widths are calibrated on the box by
tools/fetch_calib.sh (known bytes / counter per bytes-per-lane) -- LOAD_MIX below says what each kernel issues (from its
disassembly).  hbm_bytes = (factor * FETCH_SIZE + WRITE_SIZE) KB.
A launch is "live" when its duration / counter exceeds half of the kernel's maximum: the speculative CG launches that
found their solve converged return at the guard after a few microseconds and move nothing.
"""
import collections, csv, glob, json, os, re, statistics as st, sys


def short(n):
    n = n.replace('void ', '')
    m = re.search(r'k_cldr<(\w+), (\d+), (\w+)<[^>]*>, CldrSrc(\w+)<[^>]*>, (\d+), (\d+), (\d+), (\d+)', n)
    if m:
        return f"k_cldr<{m.group(1)},{m.group(2)},{m.group(3)},{m.group(4)}>"
    m = re.search(r'(k_rows|k_tile)<(\w+), (\d+), (\w+)(?:<[^>]*>)?, (\d+)', n)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)},{m.group(4)},GW{m.group(5)}{',Fold' if 'TileSrcFold' in n else ''}>"
    m = re.search(r'k_admm_lds<(\d+), (\w+), (\d+), (\w+), (\d+), (\d+), (\w+), (-?\d+)>', n)
    if m:
        return f"k_admm_lds<TPG{m.group(1)},{'band' if m.group(2) == 'true' else ('uniform' if m.group(5) != '0' else 'ragged')}{',slots' if m.group(7) == 'true' else ''},TP{m.group(8)}>"
    return n.split('(')[0][:52]


def durations(d):
    dur = collections.defaultdict(list)
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    for r in csv.DictReader(open(f)):
        dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return dur


def counters(d, name):
    out = collections.defaultdict(list)
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == name:
            out[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return out


def live_mean(v):
    m = max(v)
    lv = [x for x in v if x > 0.5 * m] if m > 0 else v
    return st.mean(lv), len(lv)


def is_spmm_in_cg(k):
    return (k.startswith('k_tile') and ('EpiLhs' in k or 'EpiStore' in k)) or (k.startswith('k_cldr') and 'EpiLhs' in k)


# bytes per lane of the global LOADS a kernel issues, as fractions of its load bytes (disassembly: the streaming kernels at
# B >= 192 read float4 per lane; k_admm_lds reads its state vectors thread-major as dwordx4 -- 8 vectors per ADMM iteration --
# and y / the first iterate as dwords -- about one vector per iteration)
LOAD_MIX = {"k_cldr": {16: 1.0}, "k_tile": {16: 1.0}, "k_rows": {16: 1.0}, "k_admm_lds": {16: 8.0 / 9.0, 4: 1.0 / 9.0}}


def fetch_factor(kernel, calib):
    mix = next((v for k, v in LOAD_MIX.items() if kernel.startswith(k)), {16: 1.0})
    f = 0.0
    for width, share in mix.items():
        c = calib.get(str(width), {}).get("fetch_bytes_per_counter_byte") if calib else None
        f += share * (c if c else (2.0 if width == 16 else 1.0))
    return f


def main():
    d, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    calib = json.load(open(sys.argv[4]))["bytes_per_lane"] if len(sys.argv) > 4 and os.path.exists(sys.argv[4]) else None
    msteps = re.search(r'--steps (\d+)', cmd)
    lds_ipl = int(msteps.group(1)) if msteps else 1          # ADMM iterations of the (single) timed k_admm_lds launch: steps <= chunk length
    dur = durations(d + '/stats')
    fetch, write = counters(d + '/fetch', 'FETCH_SIZE'), counters(d + '/write', 'WRITE_SIZE')
    tot = sum(sum(v) for v in dur.values())
    L = [f"# rocprofv3 --kernel-trace --stats -- {cmd}",
         "# default cfg2 workload (N=307, B=4096: LDS-resident path, k_admm_lds) followed by the cfg3 roofline leg",
         "# (N=10000, B=512: streaming path; cLdr inside the x / zd solves = fused k_cldr with the CG vector update folded in,",
         "# Lu and the single operators in the LDS-tiled k_tile, element-wise work in k_rows).",
         "# live = launches that did the work (duration > 50 % of the kernel's longest launch); the others are speculative CG",
         "# launches that found their solve converged and returned at the guard.  bench.py's roofline objects use live launches only.",
         f"{'kernel':52s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'pct':>6s} {'n_live':>7s} {'live_avg_us':>12s}"]
    for k in sorted(dur, key=lambda k: -sum(dur[k]))[:26]:
        v = dur[k]
        lm, ln = live_mean(v)
        L.append(f"{k:52s} {len(v):6d} {st.mean(v):10.1f} {sum(v) / 1e3:10.2f} {100 * sum(v) / tot:6.1f} {ln:7d} {lm:12.1f}")
    mix = [k for k in dur if is_spmm_in_cg(k)]
    if mix:
        lv = {k: [x for x in dur[k] if x > 0.5 * max(dur[k])] for k in mix}
        nl = sum(len(v) for v in lv.values())
        L += ["", f"# SpMM inside CG on the cfg3 leg = {', '.join(sorted(mix))}: {nl} live launches, average "
                  f"{sum(sum(v) for v in lv.values()) / nl:.1f} us -- compare bench.py's",
              "# roofline.cfg3_spmm_avg_launch_us, which times the same live launch mix with HIP events"]
    lds = [k for k in dur if k.startswith('k_admm_lds')]
    if lds:
        L += [f"# {lds[0]}: average {st.mean(dur[lds[0]]):.1f} us over {len(dur[lds[0]])} launches (bench.py roofline.avg_launch_us; the first"
              " launches of a solve run more CG iterations than the later ones)"]
    L += ["", "# PMC passes (separate runs): FETCH_SIZE / WRITE_SIZE in KB per dispatch, mean over LIVE dispatches (dispatches",
          "# skipped by the converged-CG early exit are excluded: counter > 50% of the kernel's max).",
          "# hbm_MB = (factor*FETCH_SIZE + WRITE_SIZE) KB / 1e3; factor = known bytes / counter for the kernel's load widths"
          + (" (calibrated on this box: " + ", ".join(f"{w} B/lane x{v['fetch_bytes_per_counter_byte']:.2f}" for w, v in sorted(calib.items(), key=lambda kv: int(kv[0]))) + ")" if calib else " (2 for 16 B/lane, 1 otherwise: no calibration file)"),
          f"{'kernel':44s} {'n_live':>6s} {'FETCH_KB':>10s} {'WRITE_KB':>10s} {'factor':>7s} {'hbm_MB':>10s} {'live_avg_us':>12s} {'TB/s':>7s}"]
    per = {}
    for k in sorted(fetch, key=lambda k: -sum(dur.get(k, [0]))):
        if not (k.startswith('k_rows') or k.startswith('k_tile') or k.startswith('k_admm') or k.startswith('k_cldr')):
            continue
        f, n = live_mean(fetch[k])
        w, _ = live_mean(write.get(k, [0]))
        du, _ = live_mean(dur[k]) if k in dur else (0, 0)
        ff = fetch_factor(k, calib)
        per[k] = (ff * f + w) * 1024.0
        L.append(f"{k:44s} {n:6d} {f:10.0f} {w:10.0f} {ff:7.2f} {per[k] / 1e6:10.1f} {du:12.1f} {per[k] / max(du, 1e-9) / 1e6:7.2f}")
    if os.path.isdir(d + '/fold0'):
        # the same command under MGADMM_FOLD=0: the CG vector update runs in its own kernel and the SpMM launches carry their
        # own bytes only -- cfg3 leg: T*N*B = 24 * 10 000 * 512 elements; cLdr = 2 applications (read p, write A p twice over:
        # 16 B/element), Lu = 1 application (8 B/element)
        d0 = durations(d + '/fold0')
        elems = 24 * 10000 * 512
        L += ["", "# SpMM-only launches (same command under MGADMM_FOLD=0: p = r + beta p, x += alpha p in their own k_rows<EpiPUpdate> launch);",
              "# algorithmic bytes: 16 B/element for cLdr (k_cldr: Ldr then Ldr^T in one pass), 8 B/element for Lu (k_tile), 20 B/element for the vector update",
              f"{'kernel':44s} {'n_live':>6s} {'live_avg_us':>12s} {'alg_GB':>8s} {'TB/s':>7s} {'of 8 TB/s':>10s}"]
        for k in sorted(d0, key=lambda k: -sum(d0[k])):
            bpe = 16 if (k.startswith('k_cldr') and 'EpiLhs' in k) else 8 if (k.startswith('k_tile') and 'EpiLhs' in k) else 20 if 'EpiPUpdate' in k else 0
            if not bpe:
                continue
            lm, ln = live_mean(d0[k])
            L.append(f"{k:44s} {ln:6d} {lm:12.1f} {bpe * elems / 1e9:8.3f} {bpe * elems / lm / 1e6:7.2f} {bpe * elems / lm / 1e6 / 8.0:10.2f}")
    open(out, 'w').write("\n".join(L) + "\n")
    spmm = [k for k in per if is_spmm_in_cg(k)]
    wts = {k: live_mean(fetch[k])[1] for k in spmm}
    tj = {"_source": f"{out}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs) of `{cmd}`; "
                     "hbm_bytes = (factor*FETCH_SIZE + WRITE_SIZE) KiB, factor per kernel from its load widths and the on-box calibration "
                     "(tools/fetch_calib.sh); mean over live dispatches",
          "_fetch_calibration": calib}
    lds = [k for k in per if k.startswith('k_admm_lds')]
    if lds:
        L2 = f"# {lds[0]}: the live launch runs {lds_ipl} ADMM iterations: {per[lds[0]] / lds_ipl / 1e6:.1f} MB of HBM traffic per iteration of the batch"
        open(out, 'a').write(L2 + "\n")
        tj["cfg2"] = {"kernel": lds[0], "iterations_per_profiled_launch": lds_ipl, "hbm_bytes_per_launch": per[lds[0]] / lds_ipl,
                      "_note": "hbm_bytes_per_launch is per ADMM ITERATION of the batch (bench.py multiplies by its iterations per launch)"}
    if spmm:
        tj["cfg3"] = {"kernel": "SpMM in CG = " + ", ".join(f"{k} ({wts[k]})" for k in sorted(spmm)),
                      "hbm_bytes_per_launch": sum(per[k] * wts[k] for k in spmm) / sum(wts.values()),
                      "per_kernel": {k: per[k] for k in sorted(per) if not k.startswith('k_admm')}}
    json.dump(tj, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(out))), 'traffic.json'), 'w'), indent=1)
    print("\n".join(L))


main()
