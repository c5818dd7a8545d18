"""The v4 patrol state-machine tables of the pipelined kernel's logic wave (csrc/twoarmy_engine.hip) against the literal
update they replace (twoarmy_v4.py:115-176 as restated in the sequential kernel / the oracle): the constants are read
from the source, so the test pins what is compiled."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _constants():
    src = open(glob.glob(os.path.join(ROOT, "goal-*_amd", "csrc", "twoarmy_engine.hip"))[0]).read()
    t1 = int(re.search(r"n1 = \(int\)\(\((0x[0-9A-Fa-f]+)u >> \(3 \* s1\)\)", src).group(1), 16)
    lo, hi = (int(x, 16) for x in re.search(r"s2 < 8 \? (0x[0-9A-Fa-f]+)u : (0x[0-9A-Fa-f]+)u", src).groups())
    gl, gh = (int(x, 16) for x in re.search(r"gl = \(\(\((0x[0-9A-Fa-f]+)u >> m6\).*?\n.*?gh = \(\((0x[0-9A-Fa-f]+)u >> m6\)",
                                             src, re.S).groups())
    return t1, lo, hi, gl, gh


def test_patrol_tables_equal_the_literal_update():
    t1, lo, hi, gl, gh = _constants()

    def column(y, up1):                      # twoarmy_v4.py:120-146
        y += -1 if up1 else 1
        if up1:
            if y == 3:
                up1 = 0
        elif y + 2 == 7:
            up1 = 1
        return min(max(y, 3), 5), up1, not 3 <= y <= 5

    def square(x, right):                    # twoarmy_v4.py:150-176
        x += 1 if right else -1
        if right:
            if x + 1 == 11:
                right = 0
        elif x == 5:
            right = 1
        return min(max(x, 5), 10), right, not 5 <= x <= 10

    for y in (3, 4, 5):
        for u in (0, 1):
            s1 = (y - 3) * 2 + u
            ly, lu, bad = column(y, u)
            assert (t1 >> (3 * s1)) & 7 == (ly - 3) * 2 + lu
            assert bad == (s1 in (1, 4))                     # the states the kernel sends to the sequential kernel
    for x in range(5, 11):
        for r in (0, 1):
            s2 = (x - 5) * 2 + r
            lx, lr, bad = square(x, r)
            assert ((lo if s2 < 8 else hi) >> (4 * (s2 & 7))) & 15 == (lx - 5) * 2 + lr
            assert bad == (s2 in (0, 11))
    for m6 in range(6):                      # move gates: column m6 in {0, 3} (or m4 == 2 / the draw), square m6 != 1
        assert (gl >> m6) & 1 == (1 if m6 in (0, 3) else 0)
        assert (gh >> m6) & 1 == (1 if m6 != 1 else 0)
