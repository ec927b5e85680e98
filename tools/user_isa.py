#!/usr/bin/env python3
"""tools/user_isa.py NAME [LANES ...] -- what the gfx950 compiler makes of a TRANSLATED closure's step kernel (no GPU needed).
Translates tests/js/user_models.js:NAME, writes the program csrc/amwg_core.hip would hand to hiprtc, compiles it with hipcc -S and lists
the innermost loops by VALU count (the likelihood loop's unrolled body is the largest)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "bayes.js_amd")]
import isa_audit
import user_host

name = sys.argv[1]
lanes = [int(a) for a in sys.argv[2:]] or [64]
src, arrays, meta = user_host.translated(name)
for G in lanes:
    block = min(256, meta["max_threads"])
    prog = '#include "amwg_kernel.h"\n#include "amwg_user.h"\n' + src + (
        '\nextern "C" __global__ void __launch_bounds__(%d) amwg_user_step(const amwg::StepArgs a) {\n'
        '  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];\n  amwg::step_body<amwg::UserModel, %d>(a, smem);\n}\n' % (block, G))
    d = tempfile.mkdtemp()
    f = os.path.join(d, "u.hip")
    open(f, "w").write(prog)
    out = os.path.join(d, "u.s")
    subprocess.check_call([isa_audit.HIPCC] + isa_audit.FLAGS + ["-I", isa_audit.CSRC, "-o", out, f])
    txt = open(out).read()
    meta_k = isa_audit.kernel_metadata(txt)
    print(name, "G=%d" % G, meta_k.get("amwg_user_step"), out)
    # innermost loops
    ins, labels = [], {}
    body = False
    for ln in txt.splitlines():
        if ln.startswith("amwg_user_step:"):
            body = True
            continue
        if not body:
            continue
        if ln.startswith(".Lfunc_end"):
            break
        s = ln.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(\.L\w+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if s.startswith("."):
            continue
        ins.append(s)
    loops = []
    for i, s in enumerate(ins):
        m = re.match(r"^s_c?branch\w*\s+(\.L\w+)", s)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((labels[m.group(1)], i))
    inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
    if os.environ.get("ALL_LOOPS"):
        inner = loops
    rows = []
    for a, b in inner:
        seg = ins[a:b + 1]
        c = lambda pred: sum(1 for s in seg if pred(s.split()[0]))
        rows.append((c(lambda o: o.startswith("v_")), c(lambda o: "f64" in o), c(lambda o: o.startswith("s_") and not o.startswith("s_waitcnt")),
                     c(lambda o: o.startswith("ds_")), c(lambda o: o.startswith("global_") or o.startswith("flat_")), c(lambda o: o.startswith("s_cbranch") or o.startswith("s_branch")),
                     c(lambda o: o in ("v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64")), c(lambda o: o == "v_rcp_f64_e32" or o == "v_rcp_f64_e64"), a, b))
    rows.sort(reverse=True)
    print("  innermost loops by VALU:  valu  f64  salu  ds  global  branches  div_pieces  rcp  [first..last instruction]")
    for r in rows[:6]:
        print("   ", r)
