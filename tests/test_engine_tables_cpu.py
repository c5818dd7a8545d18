"""The bit tables of the pipelined kernel's logic wave (csrc/twoarmy_engine.hip: LG_B0, LG_GL, LG_GH, LG_Y1, LG_X2)
against the literal updates they replace (twoarmy_v6.py:96-112 ball triple, twoarmy_v4.py:115-176 patrol moves, as
restated in the sequential kernel / the oracle): the constants are read from the source, so the test pins what is
compiled."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _constants():
    src = open(glob.glob(os.path.join(ROOT, "goal-*_amd", "csrc", "twoarmy_engine.hip"))[0]).read()
    return {k: int(v, 16) for k, v in re.findall(r"constexpr uint32_t (LG_[A-Z0-9]+) = (0x[0-9A-Fa-f]+)u;", src)}


def column(y, up1):                      # twoarmy_v4.py:120-146
    y += -1 if up1 else 1
    if up1:
        if y == 3:
            up1 = 0
    elif y + 2 == 7:
        up1 = 1
    return y, up1, not 3 <= y <= 5


def square(x, right):                    # twoarmy_v4.py:150-176
    x += 1 if right else -1
    if right:
        if x + 1 == 11:
            right = 0
    elif x == 5:
        right = 1
    return x, right, not 5 <= x <= 10


def test_ball_triple_and_gate_tables_over_the_12_step_phase():
    c = _constants()
    b0 = 7                                # _gen_grid: balls at x = 7, 8, 9
    for step_move in range(1, 49):        # twoarmy_v6.py:96-112: +1 for step_move % 6 in {1, 0}, -1 for {2, 3}, stay for {4, 5}
        m6 = step_move % 6
        b0 += 1 if m6 in (1, 0) else (-1 if m6 in (2, 3) else 0)
        ph2 = 2 * (step_move % 12)
        assert 6 + ((c["LG_B0"] >> ph2) & 3) == b0, step_move
        col_gate = m6 in (0, 3) or step_move % 4 == 2          # twoarmy_v4.py:117 (the drawn gate is OR-ed in by the kernel)
        sq_gate = m6 != 1                                      # twoarmy_v4.py:149
        assert (c["LG_GL"] >> ph2) & 3 == (3 if col_gate else 0), step_move
        assert (c["LG_GH"] >> ph2) & 3 == (3 if sq_gate else 0), step_move


def test_patrol_phases_walk_the_literal_bounce():
    c = _constants()
    col = [(3, 0), (4, 0), (5, 1), (4, 1)]                     # phase -> (top row, up1)
    for k, (y, up) in enumerate(col):
        assert 3 + ((c["LG_Y1"] >> (2 * k)) & 3) == y
        ny, nup, bad = column(y, up)
        assert not bad and (ny, nup) == col[(k + 1) & 3]
    off = [(y, u) for y in (3, 4, 5) for u in (0, 1) if (y, u) not in col]
    assert off == [(3, 1), (5, 0)] and all(column(y, u)[2] for y, u in off)     # sent to the sequential kernel at launch
    sq = [(5, 1), (6, 1), (7, 1), (8, 1), (9, 1), (10, 0), (9, 0), (8, 0), (7, 0), (6, 0)]
    for k, (x, r) in enumerate(sq):
        assert 5 + ((c["LG_X2"] >> (3 * k)) & 7) == x
        assert k == (x - 5 if r else 15 - x)                   # the kernel's (x, right2) -> phase at launch
        assert (k < 5) == bool(r)                              # ... and phase -> right2 at write-back
        nx, nr, bad = square(x, r)
        assert not bad and (nx, nr) == sq[(k + 1) % 10]
    off = [(x, r) for x in range(5, 11) for r in (0, 1) if (x, r) not in sq]
    assert off == [(5, 0), (10, 1)] and all(square(x, r)[2] for x, r in off)
    # spawn (twoarmy_v4.py:212-225): column at rows 4..6 keeping up1, square at x = 6 + draw keeping right2
    for up in (0, 1):
        c1x2 = 2 * col.index((3, 0) if up == 0 else (5, 1))     # inactive patrols only carry the coin
        assert col[(2 + (c1x2 & 4)) // 2] == (4, up)
    for r in (0, 1):
        for dq in range(4):
            c2x3 = 0 if r else 15
            ph = (3 + 3 * dq if c2x3 < 15 else 27 - 3 * dq) // 3
            assert sq[ph] == (6 + dq, r)
