"""VGPR / scratch table of every kernel instance in libmgadmm.so (round-3 verdict item 8): compiles each translation unit of
mixed-graph-admm_amd/csrc to gfx950 assembly with the Makefile's flags (no GPU needed) and reads the kernel descriptors.

    python tools/spill_table.py profiles/r03/kernel_registers.txt
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mixed-graph-admm_amd", "csrc")
FLAGS = ["-O3", "-fno-strict-aliasing", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I" + os.path.join(ROOT, "include"),
         "--cuda-device-only", "-S"]
TUS = {"lds_launch.hip": ["-fno-slp-vectorize"], "solver_f32.hip": [], "solver_f64.hip": [], "graph.hip": [], "build.hip": [], "frontend.hip": []}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for tu, extra in TUS.items():
            asm = os.path.join(tmp, tu + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [os.path.join(CSRC, tu), "-o", asm], check=True, capture_output=True, cwd=CSRC)
            s = open(asm).read()
            for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size: (\d+)\n(?:.*\n)*?\s+\.sgpr_spill_count: (\d+)\n(?:.*\n)*?"
                                 r"\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count: (\d+)", s):
                rows.append((tu,) + m.groups())
    dm = demangle([r[1] for r in rows])
    L = ["# kernel instances of libmgadmm.so: registers and scratch (hipcc 7.2, gfx950, flags of csrc/Makefile) -- tools/spill_table.py",
         f"{'translation unit':16s} {'vgpr':>5s} {'vgpr_spill':>10s} {'sgpr_spill':>10s} {'scratch_B':>9s}  kernel"]
    for tu, name, scratch, sspill, vgpr, vspill in sorted(rows, key=lambda r: (r[0], -int(r[5]), r[1])):
        nm = re.sub(r"\s+", " ", dm.get(name, name))
        nm = nm.split("(")[0][:150]
        L.append(f"{tu:16s} {vgpr:>5s} {vspill:>10s} {sspill:>10s} {scratch:>9s}  {nm}")
    spilling = [r for r in rows if int(r[5]) > 0]
    L.append(f"# {len(rows)} kernels, {len(spilling)} with spilled VGPRs")
    open(sys.argv[1], "w").write("\n".join(L) + "\n")
    print("\n".join(L[:3] + [l for l in L[2:] if re.search(r"\s[1-9]\d*\s+\d+\s+\d+\s+k_", l)][:60]))


main()
