"""mg_gen_obs (general MiniGrid view with occlusion, C ABI) vs the reference's recorded images / masks
(tests/golden/occlusion.npz) and vs oracle/minigrid_view_oracle.py on random batches."""
import numpy as np
import pytest
import torch

import minigrid_view_oracle as mvo
from test_minigrid_view_cpu import load_cases, load_step_cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(enc, ax, ay, d, V, see_through, carrying):
    from twoarmy_amd import minigrid_view as mv
    N, W, H, _ = enc.shape
    ty, co, st = (p.to(DEV) for p in mv.planes_from_encoded(enc))
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.int32, device=DEV)          # noqa: E731
    c = None if carrying is None else torch.tensor(np.asarray(carrying, np.uint8), device=DEV)
    img, vis = mv.gen_obs(ty, co, st, W, H, t(ax), t(ay), t(d), V, see_through, c)
    torch.cuda.synchronize()
    return img.cpu().numpy(), vis.cpu().numpy()


def test_view_kernel_matches_reference_goldens(golden_dir):
    n = 0
    for c in load_cases(golden_dir):
        carry = np.array([c["carrying"] or (0, 0, 0)], np.uint8)
        for st in (0, 1):
            img, vis = _run(c["grid"][None], [c["ax"]], [c["ay"]], [c["dir"]], c["V"], bool(st), carry)
            assert np.array_equal(img[0], c["img"][st]), (c["ci"], st)
            assert np.array_equal(vis[0], c["vis"][st]), (c["ci"], st)
        n += 1
    assert n == 30


@pytest.mark.parametrize("W,H,V,N", [(17, 17, 7, 300), (17, 17, 17, 200), (8, 30, 5, 257), (40, 40, 31, 64),
                                      (3, 3, 3, 65), (21, 9, 11, 128), (6, 6, 1, 10), (10, 10, 8, 50), (12, 7, 4, 70), (9, 9, 30, 33)])
@pytest.mark.parametrize("see_through", [False, True])
def test_view_kernel_matches_oracle_random(W, H, V, N, see_through):
    rs = np.random.RandomState(W * 1000 + H * 10 + V)
    ty = rs.choice([1, 1, 1, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 3], size=(N, W, H)).astype(np.uint8)
    co = rs.randint(0, 6, size=(N, W, H)).astype(np.uint8)
    st = np.where(ty == 4, rs.randint(0, 3, size=(N, W, H)), 0).astype(np.uint8)
    co[ty == 1] = 0
    enc = np.stack([ty, co, st], -1)
    ax, ay, d = rs.randint(0, W, N), rs.randint(0, H, N), rs.randint(0, 4, N)
    carry = np.zeros((N, 3), np.uint8)
    has = rs.rand(N) < 0.4
    carry[has] = np.stack([rs.choice([5, 6, 7], has.sum()), rs.randint(0, 6, has.sum()), np.zeros(has.sum())], -1)
    want_img, want_vis = mvo.gen_obs_batch(enc, ax, ay, d, V, see_through, carry)
    img, vis = _run(enc, ax, ay, d, V, see_through, carry)
    assert np.array_equal(vis, want_vis)
    assert np.array_equal(img, want_img)
    if not see_through and V >= 5:
        assert (want_vis == 0).sum() > 0


def test_twoarmy_engine_view_agrees_with_general_kernel():
    """Cross-check of the two view paths: the Twoarmy engine's tw_gen_obs (see-through, dir 3) equals mg_gen_obs
    run on the engine's own type / colour planes."""
    from twoarmy_amd import minigrid_view as mv
    from twoarmy_amd.engine import FIELDS, TwoarmyEngine
    eng = TwoarmyEngine(4, 128, 17, seed=9981)
    out = eng.alloc_outputs(30, obs=False, matrix=False)
    eng.rollout(30, out)
    ty, co, rec = eng.get_state()
    f = lambda name: torch.tensor(rec[:, FIELDS[name]].astype(np.int32), device=DEV)   # noqa: E731
    N = 128
    for V in (7, 17):
        want = eng.gen_obs(V).cpu().numpy()
        img, _ = mv.gen_obs(torch.tensor(ty, device=DEV), torch.tensor(co, device=DEV), None, 17, 17, f("AX"),
                            f("AY"), torch.full((N,), 3, dtype=torch.int32, device=DEV), V, True)
        assert np.array_equal(img.cpu().numpy(), want), V
    eng.close()


def test_step_kernel_matches_reference_goldens(golden_dir):
    """mg_step on the 720 transitions recorded from the reference's MiniGridEnv.step (moves, overlap rules of every
    object class, goal reward, truncation, both exception kinds)."""
    from twoarmy_amd import minigrid_view as mv
    total = 0
    for c in load_step_cases(golden_dir):
        rows = c["rows"]
        B = len(rows)
        ty, _, st = (p.to(DEV) for p in mv.planes_from_encoded(np.repeat(c["grid"][None], B, 0)))
        col = lambda k: torch.tensor(rows[:, k].astype(np.int32), device=DEV)                 # noqa: E731
        ax, ay, d, sc, a = col(0), col(1), col(2), col(3), col(4)
        r, te, tr, err = mv.step(ty, st, c["W"], c["H"], a, ax, ay, d, sc, c["max_steps"])
        got = np.stack([v.cpu().numpy().astype(np.float64) for v in (ax, ay, sc, err, te, tr, r)], 1)
        bad = np.argwhere(got != rows[:, 5:12])
        assert bad.size == 0, (c["ci"], bad[0], rows[bad[0][0]], got[bad[0][0]], float(got[bad[0][0], 6]).hex())
        total += B
    assert total == 720


def test_step_kernel_matches_oracle_random():
    from twoarmy_amd import minigrid_view as mv
    rs = np.random.RandomState(5)
    N, W, H, max_steps = 3000, 9, 6, 17
    ty = rs.choice([1, 1, 1, 2, 3, 4, 4, 5, 6, 7, 8, 9, 11], size=(N, W, H)).astype(np.uint8)
    st = np.where(ty == 4, rs.randint(0, 3, size=(N, W, H)), 0).astype(np.uint8)
    enc = np.stack([ty, np.zeros_like(ty), st], -1)
    ax, ay, d = rs.randint(0, W, N), rs.randint(0, H, N), rs.randint(0, 4, N)
    sc, a = rs.randint(0, 20, N), rs.choice([0, 1, 2, 3, 6, 4, 5, 9, -3], N)
    want = np.array([mvo.step(mvo.Grid.from_encoded(enc[n]), int(ax[n]), int(ay[n]), int(d[n]), int(sc[n]), max_steps,
                              int(a[n])) for n in range(N)], np.float64)
    t = lambda v: torch.tensor(v.astype(np.int32), device=DEV)                               # noqa: E731
    tp, _, sp = (p.to(DEV) for p in mv.planes_from_encoded(enc))
    gx, gy, gs = t(ax), t(ay), t(sc)
    r, te, tr, err = mv.step(tp, sp, W, H, t(a), gx, gy, t(d), gs, max_steps)
    got = np.stack([v.cpu().numpy().astype(np.float64) for v in (gx, gy, gs, err, te, tr, r)], 1)
    assert np.array_equal(got, want)
    assert want[:, 4].sum() > 0 and (want[:, 3] == 1).sum() > 0 and (want[:, 3] == 2).sum() > 0
