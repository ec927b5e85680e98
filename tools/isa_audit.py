#!/usr/bin/env python3
"""tools/isa_audit.py -- what the gfx950 compiler made of the step kernels (no GPU needed).

Compiles bayes.js_amd/csrc/amwg_kernels.hip (once per built-in family) to device assembly (hipcc --cuda-device-only -S, ~30 s; or reads
existing .s files with --asm) and reports per `amwg_step_kernel<Model, G, BT>` instantiation:
  * the code object metadata: VGPRs, SGPRs, vgpr/sgpr spill counts, scratch bytes, the workgroup size it was compiled for;
  * static instruction counts INSIDE LOOPS (between a label and the last backward branch to it): scratch_* (spilled VGPRs / private
    arrays), v_readlane / v_writelane (how spilled SGPRs travel), s_waitcnt, VALU -- the per-update code of a kernel is all inside
    the step loop, so anything counted here is paid per update;
and FAILS (exit 1) if a BENCHED instantiation (--gate, default: the ones bench.py times, at the workgroup class it launches them with)
   * compiled for workgroups of up to 512 threads has ANY spilled VGPR or ANY scratch_* instruction inside a loop (round 2 compiled
     everything for 1024 threads and carried 31 scratch instructions through the slot loop of the cfg4 kernel);
   * has more than --max-lane-moves v_readlane / v_writelane inside its loops.  These are how spilled SGPRs travel (a VALU move each, no
     memory): the step kernel keeps ~250 wave-uniform values alive (model constants, table offsets, masks) against ~100 scalar registers, so
     some remain by design -- the threshold is a regression guard (static count over BOTH step loops of a kernel, the ordinary one and the group-local one), not a claim of zero.
The 1024-thread class (128 VGPRs per lane) still spills VGPRs in the hierarchical kernel; it is listed, and gated on lane moves only:
bench.py measured it FASTER than the spill-free 512-thread class where the host picks it (cfg4 at 16 384 chains, cfg3), occupancy wins.

    python tools/isa_audit.py                 # compile + audit + gate
    python tools/isa_audit.py --asm /tmp/core.s --all
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bayes.js_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -falign-loops=64 -Wno-unused-function --cuda-device-only -S".split()

# the instantiations bench.py times (model, lanes per chain); any workgroup size of those is gated
# (",cert": the kernels that decide from certified values -- amwg_step_kernel_cert / amwg_sweep_kernel_cert, the defaults of cfg2 / cfg4 / cfg5; the plain
# names beside them are what options.full_evaluation = 1 / 2 runs)
DEFAULT_GATE = ["NormalModel,1,256,cert", "NormalModel,1,256", "HierNormalModel,sweep,512,cert", "HierNormalModel,sweep,512", "HierNormalModel,64,512", "HierNormalModel,32,1024",
                "PoisGlmModel,16,256,cert", "PoisGlmModel,16,256", "PoisGlmModel,64,256", "BetaBernModel,1,1024", "HierGlModel,512"]
# Spills tolerated OUTSIDE the passes, in the two certified kernels that keep 256 registers busy (two wavefronts per SIMD): the 16-lane Poisson kernel (the certified
# pass for four chains to a wavefront, the stepper, and the out-of-line expression in the reference's order) spills 6 VGPRs around its pass; the certified sweep kernel
# (the stepper, the window stream, the sweep's all-at-once decisions and the walk update by update) ~36 loop-invariant words.  No kernel may have a scratch
# instruction inside a PASS -- an innermost loop with 40 or more fp64 instructions (checked for every gated kernel below).  (Until the certified paths became kernels
# of their own -- amwg_*_kernel_cert -- both carried the lane-order expression as well: 25 and 49 spilled registers.)
# (round 6: the Normal family's certified kernel -- 64 partial sums + a block of 16 observations per lane -- parks 6 registers in accumulation registers: vgpr_spill_count
# 6 with a private segment of 0 bytes, i.e. no scratch memory; allowed as long as no scratch instruction appears)
SPILL_ALLOW = {"NormalModel,1,256,cert": {"vgpr_spill": 8, "loop_scratch": 0},
               "PoisGlmModel,16,256,cert": {"vgpr_spill": 8, "loop_scratch": 8},
               "HierNormalModel,sweep,512,cert": {"vgpr_spill": 40, "loop_scratch": 56}}


# v_readlane / v_writelane that are NOT spilled scalars: the certified pass of the Normal family broadcasts the 64 chains' means with 2 x 64 v_readlane per block of
# observations (csrc/amwg_pass.h norm_sq_pass_wave); since round 6 the rest of a pass is walked in blocks of half the length each -- 16, 8, 4, 2, 1 rounds, a loop of
# single rounds and the masked last one: seven static copies of the 128 broadcasts
LANE_MOVE_LIMITS = {"NormalModel,1,256,cert": 1500}


def compile_asm(family):
    """device assembly of the step kernels of one built-in family (amwg_kernels.hip -DAMWG_FAMILY=n)"""
    out = os.path.join(tempfile.gettempdir(), "amwg_kernels_%d.s" % family)
    subprocess.check_call([HIPCC] + FLAGS + ["-DAMWG_FAMILY=%d" % family, "-o", out, "amwg_kernels.hip"], cwd=CSRC, stderr=subprocess.DEVNULL)
    return out


def demangle(names):
    p = subprocess.run(["c++filt"] + names, capture_output=True, text=True)
    return p.stdout.splitlines() if p.returncode == 0 else names


def kernel_metadata(txt):
    m = re.search(r"amdhsa\.kernels:(.*?)amdhsa\.target", txt, re.S)
    out = {}
    for k in re.split(r"\n  - ", m.group(1)):
        name = re.search(r"\.name:\s+(\S+)", k)
        if not name:
            continue
        g = lambda f: int((re.search(r"\.%s:\s+(\d+)" % f, k) or [0, 0])[1])
        out[name.group(1)] = {"vgpr": g("vgpr_count"), "agpr": g("agpr_count"), "sgpr": g("sgpr_count"), "vgpr_spill": g("vgpr_spill_count"),
                              "sgpr_spill": g("sgpr_spill_count"), "scratch_bytes": g("private_segment_fixed_size"),
                              "max_workgroup": g("max_flat_workgroup_size"), "lds_static": g("group_segment_fixed_size")}
    return out


def kernel_bodies(txt):
    """mangled name -> list of asm lines of the function body"""
    bodies, cur, name = {}, None, None
    for line in txt.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end") or re.match(r"^\s*\.end_amdhsa_kernel", line):
                bodies[name] = cur
                cur = None
            else:
                cur.append(line)
    return bodies


def loop_stats(lines):
    """static counts inside loops: a loop = [label .. last backward branch to that label]"""
    labels, ins = {}, []
    for ln in lines:
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
    in_loop = [0] * (len(ins) + 1)
    depth_marks = []
    for i, s in enumerate(ins):
        m = re.match(r"^s_c?branch\w*\s+(\.L\w+)", s)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            depth_marks.append((labels[m.group(1)], i))
    covered = [False] * len(ins)
    for a, b in depth_marks:
        for i in range(a, b + 1):
            covered[i] = True
    cnt = {"instructions": 0, "scratch": 0, "lane_moves": 0, "waitcnt": 0, "valu": 0, "valu_f64": 0, "ds": 0, "global": 0, "smem": 0}
    tot = dict(cnt)
    for i, s in enumerate(ins):
        op = s.split()[0]
        for d in ((cnt, tot) if covered[i] else (tot,)):
            d["instructions"] += 1
            if op.startswith("scratch_") or op.startswith("buffer_") and "offen" in s and "s[0:3]" in s:
                d["scratch"] += 1
            elif op in ("v_readlane_b32", "v_writelane_b32"):
                d["lane_moves"] += 1
            elif op == "s_waitcnt":
                d["waitcnt"] += 1
            elif op.startswith("v_"):
                d["valu"] += 1
                if "f64" in op:
                    d["valu_f64"] += 1
            elif op.startswith("ds_"):
                d["ds"] += 1
            elif op.startswith("global_") or op.startswith("flat_"):
                d["global"] += 1
            elif op.startswith("s_load") or op.startswith("s_buffer_load"):
                d["smem"] += 1
    # passes: innermost loops (no other loop's backward branch inside) with >= 40 fp64 vector instructions; scratch instructions inside them
    pass_scratch, passes = 0, 0
    for a, b in depth_marks:
        if any(a2 >= a and b2 <= b and (a2, b2) != (a, b) for a2, b2 in depth_marks):
            continue
        body = ins[a:b + 1]
        reads = sum(1 for x in body if x.startswith("ds_read") or x.startswith("global_load") or x.startswith("s_load") or x.startswith("flat_load"))
        if sum(1 for x in body if x.split()[0].startswith("v_") and "f64" in x.split()[0]) >= 40 and reads >= 2:      # (a pass READS data: a loop of pure arithmetic -- the update-by-update sweep with its inlined exponential -- is not one)
            passes += 1
            pass_scratch += sum(1 for x in body if x.startswith("scratch_"))
    cnt["pass_scratch"] = pass_scratch
    cnt["passes"] = passes
    return {"in_loops": cnt, "whole_kernel": tot, "loops": len(depth_marks)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", nargs="*", help="existing device assembly files (else amwg_kernels.hip is compiled for --families)")
    ap.add_argument("--families", nargs="*", type=int, default=[0, 1, 2, 3], help="0 Normal, 1 BetaBern, 2 HierNormal, 3 PoisGlm")
    ap.add_argument("--all", action="store_true", help="print every step-kernel instantiation, not only the gated ones")
    ap.add_argument("--gate", nargs="*", default=DEFAULT_GATE)
    ap.add_argument("--max-lane-moves", type=int, default=720)      # (round 5: the sweep kernel carries three step loops -- all-at-once decisions, update by update, ordinary -- 683 static)
    ap.add_argument("--json", help="write the table here")
    args = ap.parse_args()
    if args.asm:
        asms = args.asm
    else:
        import concurrent.futures
        with concurrent.futures.ThreadPoolExecutor(4) as ex:
            asms = list(ex.map(compile_asm, args.families))
    rows, bad = [], []
    for asm in asms:
        txt = open(asm).read()
        meta, bodies = kernel_metadata(txt), kernel_bodies(txt)
        names = [n for n in meta if "amwg_step_kernel" in n or "amwg_user_step" in n or "amwg_gl_kernel" in n or "amwg_sweep_kernel" in n]
        for n, d in zip(names, demangle(names)):
            short = re.sub(r"^void amwg::amwg_(?:step|gl|sweep)_kernel(?:_cert)?<amwg::(.*)>\(.*$", r"\1", d).replace(" ", "")
            if "amwg_sweep_kernel" in d:
                short = short.replace("HierNormalModel,", "HierNormalModel,sweep,")
            if "_kernel_cert<" in d:
                short += ",cert"
            st = loop_stats(bodies.get(n, []))
            row = {"kernel": short, **meta[n], **{"loop_" + k: v for k, v in st["in_loops"].items()}, "total_instructions": st["whole_kernel"]["instructions"]}
            rows.append(row)
            gated = short in args.gate
            row["gated"] = gated
            if gated:
                why = []
                small = row["max_workgroup"] <= 512
                allow = SPILL_ALLOW.get(short, {})
                if small and row["vgpr_spill"] > allow.get("vgpr_spill", 0):
                    why.append("vgpr_spill_count %d" % row["vgpr_spill"])
                if small and row["loop_scratch"] > allow.get("loop_scratch", 0):
                    why.append("%d scratch instructions inside loops" % row["loop_scratch"])
                if small and row["loop_pass_scratch"]:
                    why.append("%d scratch instructions inside a pass (an innermost loop of fp64 arithmetic)" % row["loop_pass_scratch"])
                if row["loop_lane_moves"] > LANE_MOVE_LIMITS.get(short, args.max_lane_moves):
                    why.append("%d v_readlane/v_writelane inside loops (> %d)" % (row["loop_lane_moves"], args.max_lane_moves))
                if why:
                    bad.append((short, why))
    rows.sort(key=lambda r: r["kernel"])
    hdr = "%-34s %5s %5s %6s %6s %7s %6s | %7s %7s %7s %7s %7s" % ("kernel<Model,G,BT>", "vgpr", "sgpr", "vspill", "sspill", "scratch", "maxwg", "l.instr", "l.valu", "l.scr", "l.lane", "l.wait")
    print(hdr)
    for r in rows:
        if args.all or r["gated"]:
            print("%-34s %5d %5d %6d %6d %7d %6d | %7d %7d %7d %7d %7d%s" % (r["kernel"], r["vgpr"], r["sgpr"], r["vgpr_spill"], r["sgpr_spill"], r["scratch_bytes"], r["max_workgroup"],
                  r["loop_instructions"], r["loop_valu"], r["loop_scratch"], r["loop_lane_moves"], r["loop_waitcnt"], "  <- gated" if r["gated"] else ""))
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)
    if bad:
        for k, why in bad:
            print("FAIL %s: %s" % (k, "; ".join(why)))
        sys.exit(1)
    print("isa audit ok: %d gated instantiations: no scratch traffic inside a pass; no VGPR spill and no scratch traffic in the loops of the <= 512-thread classes "
          "except the documented allowances (%s); SGPR-spill moves within %d" % (sum(r["gated"] for r in rows), ", ".join(sorted(SPILL_ALLOW)), args.max_lane_moves))


if __name__ == "__main__":
    main()
