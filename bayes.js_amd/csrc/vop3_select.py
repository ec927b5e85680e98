#!/usr/bin/env python3
"""Build step (csrc/Makefile): re-encode the compiler's `v_cndmask_b32_e32 vD, src0, vB, vcc` as `v_cndmask_b32_e64 vD, src0, vB, vcc`
in the device assembly of the step kernels.

Why: on gfx950 the VOP2 encoding of v_cndmask_b32 (which reads VCC implicitly) issues in ~5 cycles only when it directly follows the
VALU instruction that wrote VCC; anywhere else it costs ~23 cycles -- and an fp64 select is always TWO of them after one compare, so the
second one pays (measured with tools/ubench/valu_rates.hip on an MI355X, two waves per SIMD: v_cmp + 1 select 6.6 cycles, v_cmp + 2
selects 44, the VOP3 encoding with the same vcc operand 4.9 each).  The compiler always shrinks to VOP2 when the mask is in VCC and has no
switch for it; the VOP3 form is the same operation, 4 bytes longer.

Only operands VOP3 can encode on gfx9 are touched: src0 a VGPR or an inline constant (a literal or a second scalar source would not
assemble); everything else is left as the compiler wrote it.

    vop3_select.py in.s out.s        (prints the counts)
"""
import re
import sys

INLINE = r"(?:v\d+|-?(?:[0-9]|[1-5][0-9]|6[0-4])|-1[0-6]|-?0\.5|-?1\.0|-?2\.0|-?4\.0)"
PAT = re.compile(r"^(\s*)v_cndmask_b32_e32 (v\d+), (" + INLINE + r"), (v\d+), vcc\b")


def rewrite(text):
    done = left = 0
    out = []
    for line in text.split("\n"):
        if "v_cndmask_b32_e32" in line:
            new = PAT.sub(r"\1v_cndmask_b32_e64 \2, \3, \4, vcc", line)
            if new != line:
                done += 1
            else:
                left += 1
            line = new
        out.append(line)
    return "\n".join(out), done, left


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    text, done, left = rewrite(open(src).read())
    open(dst, "w").write(text)
    print("vop3_select: %d selects re-encoded, %d left (literal or scalar src0)" % (done, left))
