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
