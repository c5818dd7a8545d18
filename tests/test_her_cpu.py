"""Buffer_gridworld.store / her_func (host logic) vs buffers recorded from the reference (tests/golden/her.npz)."""
import numpy as np


def test_her_matches_reference(golden_dir):
    from twoarmy_amd.soa.env_buffer import Buffer_gridworld
    z = np.load(golden_dir + "/her.npz")
    dt = Buffer_gridworld.ppo_dtype()
    for ci in range(int(z["n_cases"])):
        cap, seed, pre, L, cnt_before, full_before, cnt_after, full_after, end_after = z["c%d_meta" % ci]
        b = Buffer_gridworld()
        b.grid_size, b.transition, b.buffer_capacity = 17, dt, int(cap)
        b.buffer = np.zeros(int(cap), dtype=dt)
        for k in dt.names:
            b.buffer[k] = z["c%d_before_%s" % (ci, k)]
        b.counter, b.full, b.epo_counter_start = int(cnt_before), bool(full_before), int(pre)
        np.random.seed(int(seed))
        b.her_func(max_steps=50, newgoal_size_in=4)
        assert (b.counter, b.full, b.epo_counter_end) == (int(cnt_after), bool(full_after), int(end_after)), ci
        for k in dt.names:
            assert np.array_equal(b.buffer[k], z["c%d_after_%s" % (ci, k)]), (ci, k)


def test_index_record_oracle_reproduces_reference_appends(golden_dir):
    """oracle/her_oracle.py (the checker of the device HER kernel) is pinned to the reference: its index
    records, materialised as copies of the episode records with g / r / d replaced, are exactly the records
    the reference's her_func appended to its ring buffer (tests/golden/her.npz), picks replayed from the same
    np.random stream."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(golden_dir.rstrip("/")), "..", "oracle"))
    import her_oracle
    from twoarmy_amd.soa.env_buffer import Buffer_gridworld
    z = np.load(golden_dir + "/her.npz")
    names = Buffer_gridworld.ppo_dtype().names
    checked = 0
    for ci in range(int(z["n_cases"])):
        cap, seed, pre, L, cnt_before, full_before, cnt_after, full_after, end_after = (int(v) for v in z["c%d_meta" % ci])
        if cnt_before <= pre:                         # the episode itself wrapped: the reference relabels nothing useful
            continue
        before = {k: z["c%d_before_%s" % (ci, k)] for k in names}
        after = {k: z["c%d_after_%s" % (ci, k)] for k in names}
        ep = slice(pre, cnt_before)
        n_rec = cnt_before - pre
        pos = before["p"][ep][:, 4, 0:2].reshape(n_rec, 1, 2)
        term = np.zeros((n_rec, 1), np.uint8); term[-1] = 1
        # replay the picks the reference drew: np.random.choice over the first-visit indices
        fv = her_oracle.first_visit(pos[:, 0])
        np.random.seed(seed)
        chosen = np.random.choice(fv, size=min(4, fv.size), replace=False)
        choices = np.full((n_rec, 1, 4), -1, np.int32)
        choices[-1, 0, :chosen.size] = [int(np.where(fv == c)[0][0]) for c in chosen]
        rec = her_oracle.relabel(pos, term, np.zeros_like(term), np.zeros(1, np.int32),
                                 before["r"][ep].reshape(n_rec, 1), choices=choices)
        H = rec["t"].size
        assert H > 0
        dst = (cnt_before + np.arange(H)) % cap        # ring positions of the appended records
        src = pre + rec["t"]
        keep = np.array([j for j in range(H) if not (dst[j + 1:] == dst[j]).any()])   # survivors of ring overwrites
        for k in names:
            want = after[k][dst][keep]
            if k == "g":
                got = rec["goal"]
            elif k == "r":
                got = rec["reward"].reshape(-1, 1)
            elif k == "d":
                got = rec["done"].reshape(-1, 1).astype(want.dtype)
            else:
                got = before[k][src]
            assert np.array_equal(got[keep], want), (ci, k)
        assert (cnt_before + H) % cap == cnt_after % cap
        checked += 1
    assert checked >= 3


def test_store_ring_semantics():
    from twoarmy_amd.soa.env_buffer import Buffer_gridworld
    b = Buffer_gridworld()
    b.buffer_capacity = 3
    b.buffer = np.zeros(3, dtype=np.dtype([("r", np.float32, (1,))]))
    assert [b.store((np.array([float(i)]),)) for i in range(3)] == [False, False, True]
    assert b.counter == 0 and b.full
    b.store((np.array([9.0]),))
    assert b.buffer["r"][:, 0].tolist() == [9.0, 1.0, 2.0] and b.counter == 1
