"""CPU oracle (oracle/twoarmy_oracle.c) vs golden vectors recorded from the reference itself."""
import numpy as np
import pytest

import philox
import twoarmy_oracle as orc
from golden_util import SCAL, explicit_draws, load_traces

TRACES, SEED = load_traces()


def test_philox_c_matches_numpy():
    rs = np.random.RandomState(0)
    for _ in range(50):
        seed = int(rs.randint(0, 2**31)) | (int(rs.randint(0, 2**31)) << 32)
        eid, t, slot = int(rs.randint(0, 2**31)), int(rs.randint(0, 2**31)), int(rs.randint(0, 12))
        assert orc.lib().tw_oracle_draw_word(seed, eid, t, slot) == int(philox.draw_word(seed, eid, t, slot))


def test_philox_known_answer():
    # Random123 KAT: philox4x32-10, counter = key = 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    out = philox.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in out] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    out = philox.philox4x32_10(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(x) for x in out] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]


@pytest.mark.parametrize("idx", range(len(TRACES)))
def test_trace(idx):
    tr = TRACES[idx]
    env = orc.OracleEnv(int(tr["variant"]), SEED, int(tr["env_id"]))
    nat = explicit_draws(tr) if int(tr["natural"]) else None
    t = 0
    for k, op in enumerate(tr["op"]):
        ctx = "%s op#%d=%d" % (tr["name"], k, op)
        if op == -1:
            obs = env.reset()
            assert np.array_equal(obs, tr["obs"][k]), ctx
        else:
            draws = None if nat is None else nat.get(t, np.zeros(8, np.uint32))
            t += 1
            obs, r, te, trn, err = env.step(int(op), draws)
            code = {None: 0, "AttributeError": 1, "AssertionError": 2, "TypeError": 3}[err]
            assert code == int(tr["err"][k]), ctx
            if code == 0:
                assert np.array_equal(obs, tr["obs"][k]), ctx
                assert r == float(tr["reward"][k]), ctx          # exact: same float64 literals
                assert te == bool(tr["term"][k]) and trn == bool(tr["trunc"][k]), ctx
        f = env.flags()
        assert f["agent_pos"] == tuple(tr["agent"][k]), ctx
        for j, name in enumerate(SCAL[:-1]):
            assert int(f[name]) == int(tr["scal"][k][j]), ctx + " " + name
        assert env.e.dir == int(tr["scal"][k][10]), ctx
        assert tuple(zip(f["ball_x"], f["ball_y"])) == tuple(map(tuple, tr["balls"][k])), ctx
        if env.e.o1_valid:
            assert [(env.e.o1_x[i], env.e.o1_y[i]) for i in range(3)] == list(map(tuple, tr["o1"][k])), ctx
        else:
            assert (tr["o1"][k] == -1).all(), ctx
        if env.e.o2_valid:
            assert [(env.e.o2_x[i], env.e.o2_y[i]) for i in range(4)] == list(map(tuple, tr["o2"][k])), ctx
        else:
            assert (tr["o2"][k] == -1).all(), ctx
        assert np.array_equal(env.grid_encode(), tr["grid"][k]), ctx
        assert np.array_equal(env.matrix(), tr["matrix"][k].astype(np.float32)), ctx
        assert tr["pos"][k].tolist() == [env.e.ay, env.e.ax, env.e.goal_y, env.e.goal_x], ctx


def test_views(golden_dir):
    z = np.load(golden_dir + "/views.npz")
    env = orc.OracleEnv(4)
    for c in range(len(z["dir"])):
        g = z["grid"][c]                       # [x][y][c]
        for ch, name in enumerate(("type", "colour", "state")):
            plane = np.ascontiguousarray(g[:, :, ch].T).reshape(-1)
            getattr(env.e, name)[:] = (type(getattr(env.e, name)))(*plane.tolist())
        env.e.ax, env.e.ay = int(z["agent"][c][0]), int(z["agent"][c][1])
        env.e.dir = int(z["dir"][c])
        for V in z["view_sizes"]:
            img = env.gen_obs(view=int(V))
            assert np.array_equal(img, z["img_%03d_V%d" % (c, V)]), (c, V)


def test_known_answers_v6():
    """SURVEY.md section 4, K0/K1/K3 spot values (redundant with the traces; readable pins)."""
    env = orc.OracleEnv(6)
    xs = []
    for _ in range(6):
        env.step(6)
        xs.append(env.flags()["ball_x"])
    assert xs == [(8, 9, 10), (7, 8, 9), (6, 7, 8), (6, 7, 8), (6, 7, 8), (7, 8, 9)]
    env = orc.OracleEnv(6)
    obs, r, te, tr, _ = env.step(1)
    assert (r, te, tr) == (-0.01, False, False) and env.flags()["pone"]
    g = env.grid_encode()
    for (x, y) in [(4, 11), (5, 11), (4, 12), (5, 12), (8, 11), (9, 11), (8, 12), (9, 12)]:
        assert tuple(g[x, y]) == (2, 5, 0)
    assert tuple(obs[4 - 4 + 8, 11 - 15 + 16]) == (1, 0, 0)      # not yet visible in the t=1 obs
    obs, *_ = env.step(6)
    assert tuple(obs[4 - 4 + 8, 11 - 15 + 16]) == (2, 5, 0)      # visible from t=2
