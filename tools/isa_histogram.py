#!/usr/bin/env python3
"""Instruction classes of the trace kernel's bounce loop, by phase, weighted by the phases' measured time shares.

The PT_STAMPS=1 diagnostic build brackets the loop's phases with s_memtime (csrc/pt_kernels.hip: t0 loop top, t1 after regeneration,
ta / tb around pass 1 of the closest-hit search, t2 after pass 2, t3 after shading); tools/stamps.py measures the phases' shares of a
wave-bounce on the GPU.  This script compiles that build to assembly (no GPU needed), cuts the loop of pt_trace_kernel<true,1,3>
at the stamps, classifies every instruction between two stamps, and prints per class
    sum over phases of  (time share of the phase) x (class's share of the phase's VALU instructions)
i.e. an estimate of the class's share of the loop's VALU issue -- exact if a phase's instructions all ran equally often (inner loops
make it an estimate; the phases are cut so that each is dominated by one loop body).
usage: python tools/isa_histogram.py [regen%% pass1%% pass2%% shade%%]   (default: the shares of profiles/r04/stamps_r04.txt if present)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "oclpathtracer_amd", "csrc")
ASM = "/tmp/pt_kernels_stamps.s"
KERNEL = "_Z15pt_trace_kernelILb1ELi1ELi3EEv13PtTraceParams"

def build():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
           "-DPT_STAMPS=1", "--cuda-device-only", "-S", "-o", ASM, "pt_kernels.hip"]
    subprocess.run(cmd, cwd=SRC, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

def classify(op, args):
    if op.startswith("v_pk_"): return "packed f32 (v_pk_*)"
    if op.endswith("_f64") or "_f64_" in op: return "binary64"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", op): return "transcendental (quarter rate)"
    if op.startswith("v_cvt_"): return "convert"
    if re.match(r"v_(cmp|cmpx)_", op) or op.startswith("v_cndmask") or op.startswith("v_addc_co") or op.startswith("v_subb"): return "compare / select / carry"
    if re.match(r"v_(min|max|med3)", op): return "min / max / med3"
    if re.match(r"v_(fma|fmac|fmaak|fmamk|mul|add|sub|mad|mac)_(f32|legacy_f32)", op) or op in ("v_subrev_f32",):
        scalar = bool(re.search(r"(?<![a-z0-9_])(s\d+|s\[\d+:\d+\]|vcc|0x[0-9a-f]+|-?\d+\.\d+|-?\d+(?![:\]]))", args.split(",", 1)[1] if "," in args else ""))
        return "FMA / MUL / ADD f32, an SGPR or constant operand" if scalar else "FMA / MUL / ADD f32, all VGPR"
    if re.match(r"v_(mov|readlane|readfirstlane|writelane|swap|permlane|accvgpr)", op): return "move / lane access"
    if op.startswith("v_"): return "integer / bit (RNG, indices, masks)"
    if op.startswith("ds_"): return "LDS"
    if re.match(r"(global|buffer|scratch|flat)_", op): return "vector memory"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime"): return "scalar memory"
    if op.startswith("s_"): return "SALU / branch / wait"
    return "other"

def main():
    build()
    t = open(ASM).read()
    a = t.index(KERNEL + ":"); b = t.index(".Lfunc_end", a)
    lines = t[a:b].split("\n")
    stamps = [i for i in range(len(lines)) if "s_memtime" in lines[i]]   # t0 t1 ta tb t2 t3, in program order (the kernel has no others)
    names = ["regeneration (t0-t1)", "search set-up (t1-ta)", "pass 1 (ta-tb)", "pass 2 (tb-t2)", "shading (t2-t3)"]
    if len(stamps) != 6:
        print("expected 6 stamps in the kernel, found %d at %r" % (len(stamps), stamps)); sys.exit(1)
    args = [float(x) for x in sys.argv[1:5]]
    if len(args) != 4:
        f = os.path.join(ROOT, "profiles", "r04", "stamps_r04.txt")
        m = re.search(r"shares: regen ([\d.]+) pass1 ([\d.]+) pass2 ([\d.]+) shade ([\d.]+)", open(f).read()) if os.path.exists(f) else None
        args = [float(x) for x in m.groups()] if m else [6.5, 29.5, 30.7, 33.3]
    regen, p1, p2, shade = args
    share = {names[0]: regen, names[1]: 0.0, names[2]: p1, names[3]: p2, names[4]: shade}   # (the set-up is inside the "loop" figure of stamps.py: counted with pass 2)
    hist, per_phase = {}, {}
    for k, name in enumerate(names):
        counts = {}
        for l in lines[stamps[k] + 1:stamps[k + 1]]:
            m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)(?:\s*;.*)?$", l)
            if not m or m.group(1).startswith("."): continue
            c = classify(m.group(1), m.group(2))
            counts[c] = counts.get(c, 0) + 1
        per_phase[name] = counts
    # the set-up's instructions run once per bounce like pass 2's head: fold them into pass 2
    for c, n in per_phase[names[1]].items(): per_phase[names[3]][c] = per_phase[names[3]].get(c, 0) + n
    valu = lambda c: not c in ("LDS", "vector memory", "scalar memory", "SALU / branch / wait", "other")
    print("static instructions of the bounce loop of %s (PT_STAMPS=1 build), by phase; shares of a wave-bounce: regeneration %.1f %%, pass 1 %.1f %%, pass 2 %.1f %%, shading %.1f %%"
          % ("pt_trace_kernel<true,1,3>", regen, p1, p2, shade))
    for name in (names[0], names[2], names[3], names[4]):
        counts = per_phase[name]
        nv = sum(n for c, n in counts.items() if valu(c))
        print("\n%s: %d instructions, %d of them VALU" % (name, sum(counts.values()), nv))
        for c, n in sorted(counts.items(), key=lambda x: -x[1]):
            print("   %-52s %4d  %s" % (c, n, "%5.1f %% of the phase's VALU" % (100.0 * n / nv) if valu(c) else ""))
            if valu(c): hist[c] = hist.get(c, 0.0) + share[name] * n / nv
    tot = sum(hist.values())
    print("\nestimated share of the loop's VALU issue by class (time share of the phase x the class's share of its VALU instructions):")
    for c, v in sorted(hist.items(), key=lambda x: -x[1]):
        print("   %-52s %5.1f %%" % (c, 100.0 * v / tot))

if __name__ == "__main__":
    main()
