"""oracle/minigrid_view_oracle.py (checker of the general MiniGrid view kernel) vs images and visibility masks the
reference's own gen_obs / gen_obs_grid produced (tests/golden/occlusion.npz)."""
import numpy as np

import minigrid_view_oracle as mvo


def load_cases(golden_dir):
    z = np.load(golden_dir + "/occlusion.npz")
    for ci in range(int(z["n_cases"])):
        W, H, ax, ay, d, V, has_carry, ct, cc, cs = (int(v) for v in z["c%03d_meta" % ci])
        yield dict(ci=ci, W=W, H=H, ax=ax, ay=ay, dir=d, V=V, carrying=(ct, cc, cs) if has_carry else None,
                   grid=z["c%03d_grid" % ci], img=[z["c%03d_img%d" % (ci, s)] for s in (0, 1)],
                   vis=[z["c%03d_vis%d" % (ci, s)] for s in (0, 1)])


def test_view_oracle_matches_reference(golden_dir):
    n = occluded = 0
    for c in load_cases(golden_dir):
        assert c["grid"].shape == (c["W"], c["H"], 3)
        for st in (0, 1):
            img, vis = mvo.gen_obs(c["grid"], c["ax"], c["ay"], c["dir"], c["V"], bool(st), c["carrying"])
            assert np.array_equal(img, c["img"][st]), (c["ci"], st)
            assert np.array_equal(vis.astype(np.uint8), c["vis"][st]), (c["ci"], st)
        occluded += int((c["vis"][0] == 0).sum())
        n += 1
    assert n == 30 and occluded > 100          # the occlusion path is really exercised


def load_step_cases(golden_dir):
    z = np.load(golden_dir + "/mgstep.npz")
    for ci in range(int(z["n_cases"])):
        W, H, max_steps = (int(v) for v in z["c%02d_meta" % ci])
        yield dict(ci=ci, W=W, H=H, max_steps=max_steps, grid=z["c%02d_grid" % ci], rows=z["c%02d_rows" % ci])


def test_step_oracle_matches_reference(golden_dir):
    """rows: ax, ay, dir, step_count before | action | ax, ay, step_count after, error, terminated, truncated, reward."""
    n = 0
    for c in load_step_cases(golden_dir):
        world = mvo.Grid.from_encoded(c["grid"])
        for row in c["rows"]:
            ax, ay, d, sc, a = (int(v) for v in row[:5])
            got = mvo.step(world, ax, ay, d, sc, c["max_steps"], a)
            want = tuple(int(v) for v in row[5:11]) + (float(row[11]),)
            assert got[:4] == want[:4] and (int(got[4]), int(got[5])) == want[4:6] and got[6] == want[6], (c["ci"], row)
            n += 1
    assert n == 720


def test_add_carry_flood_identity():
    """The row sweeps of process_vis as mg_gen_obs computes them (csrc/minigrid_view.hip: flood_right / flood_left):
    r[i] = m[i] | (r[i-1] & q[i]) over the 64-bit row mask of all envs of a wavefront equals (((R + m) ^ R) & R) | m with
    R = q | m, and the mirror sweep is the same thing between two bit reversals -- checked against the definition on
    random masks with the kernel's segment structure (E = 64 // V envs of V bits, sweeps cut at env boundaries)."""
    import random
    M = (1 << 64) - 1

    def add_right(m, q):
        R = q | m
        return ((((R + m) & M) ^ R) & R) | m

    def brev(x):
        return int(format(x, "064b")[::-1], 2)

    def def_right(m, q):
        r = 0
        for i in range(64):
            b = (m >> i) & 1
            if i > 0 and (r >> (i - 1)) & 1 and (q >> i) & 1:
                b = 1
            r |= b << i
        return r

    def def_left(m, q):
        r = 0
        for i in range(63, -1, -1):
            b = (m >> i) & 1
            if i < 63 and (r >> (i + 1)) & 1 and (q >> i) & 1:
                b = 1
            r |= b << i
        return r

    rnd = random.Random(7)
    for V in (3, 5, 7, 9, 17, 31):
        E = 64 // V
        allm = (1 << (E * V)) - 1
        seg_first = sum(1 << (e * V) for e in range(E))
        seg_last = (seg_first << (V - 1)) & M
        for _ in range(3000):
            p = rnd.getrandbits(64) & allm if rnd.random() < 0.6 else (rnd.getrandbits(64) | rnd.getrandbits(64)) & allm
            m = rnd.getrandbits(64) & rnd.getrandbits(64) & allm
            q = (p << 1) & ~seg_first & allm
            assert add_right(m, q) == def_right(m, q)
            q2 = (p >> 1) & ~seg_last & allm
            assert brev(add_right(brev(m), brev(q2))) == def_left(m, q2)
