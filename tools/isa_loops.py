#!/usr/bin/env python3
"""tools/isa_loops.py -- a per-loop breakdown of one step kernel's gfx950 assembly (no GPU needed): every backward branch delimits a loop; for each, the
static instruction count and what it is made of (vector / fp64 / scalar / v_readlane+v_writelane / LDS / scratch / s_waitcnt).  tools/isa_audit.py gates the
totals; this lists WHERE a kernel's instructions sit -- e.g. the nine inlined copies of the rnorm loop in the certified sweep kernel's step loop (DESIGN.md
section 7: what a per-model specialised stepper would remove).

    python tools/isa_loops.py "amwg_sweep_kernel_cert<amwg::HierNormalModel, 512>" [--family 2] [--min 40] > profiles/r05_isa_loops_sweep_cert.txt
"""
import argparse
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_audit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel", help="substring of the demangled kernel name")
    ap.add_argument("--family", type=int, default=2, help="0 Normal, 1 beta-Bernoulli, 2 hierarchical, 3 Poisson (amwg_kernels.hip -DAMWG_FAMILY)")
    ap.add_argument("--asm", help="an existing .s file instead of compiling")
    ap.add_argument("--min", type=int, default=40, help="list loops of at least this many instructions")
    args = ap.parse_args()
    asm = args.asm or isa_audit.compile_asm(args.family)
    txt = open(asm).read()
    names = sorted(set(re.findall(r"^(_Z\w+):", txt, flags=re.M)))
    dem = dict(zip(names, isa_audit.demangle(names)))
    want = args.kernel.replace(" ", "")
    hits = [n for n in names if want in dem[n].replace(" ", "") and "StepArgs" in dem[n]]
    if len(hits) != 1:
        sys.exit("kernel %r matches %d symbols: %s" % (args.kernel, len(hits), [dem[h] for h in hits][:8]))
    name = hits[0]
    i = txt.index(name + ":")
    body = txt[i:txt.index(".Lfunc_end", i)].splitlines()
    labels, ins = {}, []
    for ln in body:
        s = ln.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        ins.append(s)
    loops = []
    for k, s in enumerate(ins):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", s)
        if m and m.group(1) in labels and labels[m.group(1)] <= k:
            loops.append((labels[m.group(1)], k))
    loops.sort()
    print("%s\n%d instructions, %d loops (backward branches); loops of >= %d instructions:" % (dem[name], len(ins), len(loops), args.min))
    print("%7s %7s %6s | %6s %5s %6s %5s %4s %7s %5s" % ("first", "last", "instr", "vector", "fp64", "scalar", "lane", "lds", "scratch", "wait"))
    for a, b in loops:
        seg = ins[a:b + 1]
        if len(seg) < args.min:
            continue
        op = [s.split()[0] for s in seg]
        print("%7d %7d %6d | %6d %5d %6d %5d %4d %7d %5d" % (
            a, b, len(seg), sum(o.startswith("v_") for o in op), sum("_f64" in o for o in op),
            sum(o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop")) for o in op), sum(o.startswith(("v_readlane", "v_writelane")) for o in op),
            sum(o.startswith("ds_") for o in op), sum(o.startswith(("scratch_", "buffer_")) for o in op), sum(o.startswith("s_waitcnt") for o in op)))


if __name__ == "__main__":
    main()
