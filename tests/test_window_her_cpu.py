"""Hindsight relabelling of the 9-frame window records (predictor / self-orientation entry points) vs buffers recorded
from the reference's own pre_her_func / pre_f_her_func (tests/golden/window_her.npz, oracle/gen_golden.py window_her):
the host mirror, and the index-record formulation the device kernel implements (oracle/her_oracle.py with skip = 4)."""
import numpy as np

import her_oracle


def _cases(golden_dir):
    z = np.load(golden_dir + "/window_her.npz")
    for ci in range(int(z["n_cases"])):
        meta = [int(v) for v in z["c%d_meta" % ci]]
        names = ["s", "a", "p", "g", "r", "d", "a_logp"] + (["f"] if meta[9] else [])
        yield ci, meta, names, {k: z["c%d_before_%s" % (ci, k)] for k in names}, {k: z["c%d_after_%s" % (ci, k)] for k in names}


def test_window_her_matches_reference(golden_dir):
    from twoarmy_amd.soa.env_buffer import Buffer_gridworld
    n = 0
    for ci, meta, names, before, after in _cases(golden_dir):
        cap, seed, pre, L, cnt_before, full_before, cnt_after, full_after, end_after, with_f = meta
        b = Buffer_gridworld()
        dt = Buffer_gridworld.window_dtype(17, bool(with_f))
        assert list(dt.names) == names
        b.grid_size, b.pre_transition, b.buffer_pre_capacity = 17, dt, cap
        b.pre_buffer = np.zeros(cap, dtype=dt)
        for k in names:
            b.pre_buffer[k] = before[k]
        b.pre_counter, b.pre_full, b.epo_counter_start = cnt_before, bool(full_before), pre
        np.random.seed(seed)
        (b.pre_f_her_func if with_f else b.pre_her_func)(max_steps=50, newgoal_size_in=4)
        assert (b.pre_counter, b.pre_full, b.epo_counter_end) == (cnt_after, bool(full_after), end_after), ci
        for k in names:
            assert np.array_equal(b.pre_buffer[k], after[k]), (ci, k)
        n += 1
    assert n == 6


def test_pre_store_ring_semantics():
    from twoarmy_amd.soa.env_buffer import Buffer_gridworld
    for store, cnt, full, ring in (("pre_store", "pre_counter", "pre_full", "pre_buffer"),
                                   ("future_3_position_store", "fp_counter", "fp_full", "fp_buffer")):
        b = Buffer_gridworld()
        b.buffer_pre_capacity = 3
        setattr(b, ring, np.zeros(3, dtype=np.dtype([("r", np.float32, (1,))])))
        flags = [getattr(b, store)((np.array([float(i)]),)) for i in range(5)]
        assert flags == [False, False, True, True, True]
        assert getattr(b, cnt) == 2 and getattr(b, full) and getattr(b, ring)["r"][:, 0].tolist() == [3.0, 4.0, 2.0]


def window_records_from_index_records(rec, S, P, A, R, D, LP, F=None):
    """Materialise index records (t, goal, reward, done) as the 9-frame windows the reference would hold for them.
    S / P: frame and position of state m (m = 0 the reset state); A, R, D, LP, F: per-step values.  A relabelled
    trajectory is the run of records up to the one with done = 1 (transition J): the window of transition e shows the
    states e-3 .. e+5 clipped to [0, J+1] and the transitions e .. e+4 clipped to J."""
    H = rec["t"].size
    out = dict(s=np.empty((H, 9, S.shape[1])), p=np.empty((H, 9, 2)), a=np.empty((H, 5, 1), np.int64), r=np.empty((H, 5, 1)),
               d=np.empty((H, 5, 1), np.int64), a_logp=np.empty((H, 5, 1)), g=rec["goal"].astype(np.float64))
    if F is not None:
        out["f"] = np.empty((H, 5, 2))
    ends = np.flatnonzero(rec["done"])
    j0 = 0
    for je in ends:
        J = int(rec["t"][je])
        for j in range(j0, je + 1):
            e = int(rec["t"][j])
            st = np.clip(np.arange(e - 3, e + 6), 0, J + 1)
            tr = np.minimum(np.arange(e, e + 5), J)
            out["s"][j], out["p"][j] = S[st], P[st]
            out["a"][j, :, 0], out["a_logp"][j, :, 0] = A[tr], LP[tr]
            out["r"][j, :, 0] = np.where(tr == J, np.float32(0.9), R[tr])
            out["d"][j, :, 0] = np.where(tr == J, 1, D[tr])
            if F is not None:
                out["f"][j] = F[tr]
        j0 = je + 1
    return out


def episode_arrays(before, pre, n_rec, with_f):
    """Per-state / per-step arrays of the stored episode: record i's frames 0..4 are the states i-3 .. i+1 and its slot 0
    is transition i."""
    ep = slice(pre, pre + n_rec)
    S = np.concatenate([before["s"][pre][0:1], before["s"][ep][:, 4]]).astype(np.float64)
    P = np.concatenate([before["p"][pre][0:1], before["p"][ep][:, 4]])
    A, R, D, LP = (before[k][ep][:, 0, 0] for k in ("a", "r", "d", "a_logp"))
    return S, P, A, R, D, LP, (before["f"][ep][:, 0] if with_f else None)


def test_index_records_with_skip_4_reproduce_the_reference_window_appends(golden_dir):
    """What VecPPOTrainer / VecSoATrainer train on for these agents -- index records from the relabelling with skip = 4,
    stacks gathered from one frame per step -- is record for record what the reference appended to its window buffer:
    all nine frames, positions, the five action / reward / done / log-prob (/ f) slots and the goal."""
    checked = 0
    for ci, meta, names, before, after in _cases(golden_dir):
        cap, seed, pre, L, cnt_before, full_before, cnt_after, full_after, end_after, with_f = meta
        if cnt_before <= pre:
            continue
        n_rec = cnt_before - pre
        assert n_rec == L                                  # one window per transition (episodes longer than 4 steps)
        S, P, A, R, D, LP, F = episode_arrays(before, pre, n_rec, with_f)
        assert np.array_equal(before["p"][pre + n_rec - 5][8], P[L])      # newest frame of record i = state i + 5
        pos = P[1:].reshape(L, 1, 2).astype(np.float32)
        term = np.zeros((L, 1), np.uint8); term[-1] = 1
        # the picks the reference drew: np.random.choice over the first-visit indices of the WINDOW sequence p[:, 8]
        _, fv_ref = np.unique(before["p"][pre:cnt_before][:, 8, 0:2], return_index=True, axis=0)
        fv = her_oracle.first_visit(pos[4:, 0])
        assert np.array_equal(fv, fv_ref)                  # the four terminal repeats add no first visit
        np.random.seed(seed)
        chosen = np.random.choice(fv_ref, size=min(4, fv_ref.size), replace=False)
        choices = np.full((L, 1, 4), -1, np.int32)
        choices[-1, 0, :chosen.size] = [int(np.where(fv == c)[0][0]) for c in chosen]
        rec = her_oracle.relabel(pos, term, np.zeros_like(term), np.zeros(1, np.int32), R.reshape(L, 1).astype(np.float32),
                                 choices=choices, skip=4)
        H = rec["t"].size
        assert H > 0
        want_w = window_records_from_index_records(rec, S, P, A, R, D, LP, F)
        dst = (cnt_before + np.arange(H)) % cap
        keep = np.array([j for j in range(H) if not (dst[j + 1:] == dst[j]).any()])
        for k in names:                                    # PPO_Predictor.update reads the buffer as float32 (:126-131)
            cmp_t = np.int64 if k in ("a", "d") else np.float32
            assert np.array_equal(want_w[k][keep].astype(cmp_t), after[k][dst][keep].astype(cmp_t)), (ci, k)
        assert (cnt_before + H) % cap == cnt_after % cap
        checked += 1
    assert checked >= 4
