#!/usr/bin/env python3
"""Instruction count of the common path of the pipelined kernel's logic loop (tw_pipe_kernel, wave 0) per
instantiation, from the gfx950 ISA (`make -C .../csrc asm`).  The logic wave is issue-bound (one instruction per
~4.3 cycles whatever its type), so this count x 4.3 cycles is the per-step chain length (engine section)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "goal-conditioned-reinforcement-learning-with-environmental-and-policy-priors_amd", "csrc")


def main():
    if "--no-build" not in sys.argv:
        subprocess.check_call(["make", "-s", "-C", CSRC, "asm"], stderr=subprocess.DEVNULL)
    s = open(os.path.join(CSRC, "twoarmy_engine.s")).read()
    for m in re.finditer(r"^(_ZN\S*tw_pipe_kernelILi(\d)ELi(\d+)ELi1E\S*):", s, re.M):
        a, b = m.end(), s.index(".Lfunc_end", m.end())
        body = s[a:b].split("\n")
        seg = [l for l in body if l.strip() and not l.strip().startswith((";", ".loc", ".cfi", ".p2align"))]
        # the step loop = the label-to-back-branch span around the record's shift-OR chain (its last link shifts by 25)
        mark = next(i for i, l in enumerate(seg) if "v_lshl_or_b32" in l and ", 25, " in l)
        pos = {l.split(":")[0]: i for i, l in enumerate(seg) if l.startswith(".LBB")}
        best = None
        for i, l in enumerate(seg):
            t = l.split()
            if len(t) >= 2 and t[0].startswith(("s_branch", "s_cbranch")) and t[1] in pos and pos[t[1]] <= mark <= i:
                if best is None:                 # the first backward branch behind the mark closes the step loop; later
                    best = (pos[t[1]], i)        # ones are the out-of-line rare blocks jumping back into it
        loop = [l for l in seg[best[0]:best[1] + 1] if not l.startswith(".LBB")]
        ins = [l.split()[0] for l in loop]
        kinds = collections.Counter("salu" if i.startswith("s_") else "lds" if i.startswith("ds_") else "valu" for i in ins)
        print("v%s PG=%-2s  %3d instructions on the logic loop's common path %s" % (m.group(2), m.group(3), len(ins), dict(kinds)))
        i0, i1 = best
        body = seg
        if "--dump" in sys.argv and m.group(2) == "4" and m.group(3) == "4":
            open("/tmp/v4pg4_logic.s", "w").write("\n".join(body[i0:i1]))


if __name__ == "__main__":
    main()
