#!/usr/bin/env python3
"""Instruction count of the pipelined kernel's logic loop (between the s_setprio pair of tw_pipe_kernel) per
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
    for m in re.finditer(r"^(_ZN\S*tw_pipe_kernelILi(\d)ELi(\d+)E\S*):", s, re.M):
        a, b = m.end(), s.index(".Lfunc_end", m.end())
        body = s[a:b].split("\n")
        i0 = next(i for i, l in enumerate(body) if "s_setprio 3" in l)
        i1 = next(i for i, l in enumerate(body) if "s_setprio 0" in l)
        ins = [l.split()[0] for l in body[i0:i1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        kinds = collections.Counter("salu" if i.startswith("s_") else "lds" if i.startswith("ds_") else "valu" for i in ins)
        # the loop's common path: from its header label to the back-edge (rare-event blocks are laid out behind it)
        seg = [l for l in body[i0:i1] if l.strip() and not l.strip().startswith((";", ".loc", ".cfi"))]
        lab = [i for i, l in enumerate(seg) if l.startswith(".LBB")]
        back = [i for i, l in enumerate(seg) if l.strip().startswith("s_branch")]
        hot = len([l for l in seg[lab[0]:back[0] + 1] if not l.startswith(".LBB")]) if lab and back else -1
        print("v%s PG=%-2s  %3d instructions between the s_setprio pair %s, %3d on the loop's common path"
              % (m.group(2), m.group(3), len(ins), dict(kinds), hot))
        if "--dump" in sys.argv and m.group(2) == "4" and m.group(3) == "4":
            open("/tmp/v4pg4_logic.s", "w").write("\n".join(body[i0:i1]))


if __name__ == "__main__":
    main()
