"""HIP engine (through the C ABI) vs golden vectors from the reference and vs the CPU oracle."""
import numpy as np
import pytest
import torch

import twoarmy_oracle as orc
from golden_util import SCAL, explicit_draws, load_traces

pytestmark = pytest.mark.gpu

TRACES, SEED = load_traces()
F = None


def _engine(*a, **k):
    from twoarmy_amd.engine import TwoarmyEngine
    return TwoarmyEngine(*a, **k)


def padded_rows(t, nlead, width):
    from twoarmy_amd.engine import padded_rows as f
    return f(t, nlead, width)


def _fields():
    from twoarmy_amd._lib import FIELDS
    return FIELDS


def _grid_encode(ty, co):
    t = ty.reshape(17, 17)
    c = co.reshape(17, 17)
    return np.stack([t.T, c.T, np.zeros_like(t.T)], axis=-1)


@pytest.mark.parametrize("idx", range(len(TRACES)))
def test_trace_vs_reference_golden(idx):
    """Every recorded reference trace replayed one tw_step at a time (N=1, no auto-reset)."""
    tr = TRACES[idx]
    Fd = _fields()
    eng = _engine(int(tr["variant"]), 1, 17, seed=SEED, env_id0=int(tr["env_id"]))
    out = eng.alloc_outputs()
    nat = explicit_draws(tr) if int(tr["natural"]) else None
    act = torch.zeros(1, dtype=torch.int32, device="cuda")
    t = 0
    for k, op in enumerate(tr["op"]):
        ctx = "%s op#%d=%d" % (tr["name"], k, op)
        if op == -1:
            obs = torch.empty((1, 17, 17, 3), dtype=torch.uint8, device="cuda")
            eng.reset(obs=obs)
            assert np.array_equal(obs.cpu().numpy()[0], tr["obs"][k]), ctx
            ty, co, rec = eng.get_state()
        else:
            act[0] = int(op)
            draws = None
            if nat is not None:
                w = nat.get(t, np.zeros(8, np.uint32))
                draws = torch.from_numpy(w.view(np.int32).copy()).cuda().reshape(1, 8)
            t += 1
            eng.step(act, out, draws=draws)
            ty, co, rec = eng.get_state()
            err = int(rec[0, Fd["ERROR"]])
            assert err == int(tr["err"][k]), ctx
            if err == 0:
                assert np.array_equal(out["obs"].cpu().numpy()[0], tr["obs"][k]), ctx
                assert out["reward"].item() == np.float32(tr["reward"][k]), ctx
                assert bool(out["terminated"].item()) == bool(tr["term"][k]), ctx
                assert bool(out["truncated"].item()) == bool(tr["trunc"][k]), ctx
                assert np.array_equal(out["matrix"].cpu().numpy()[0], tr["matrix"][k].astype(np.float32)), ctx
                assert out["pos"].cpu().numpy()[0].tolist() == tr["pos"][k][:2].tolist(), ctx
        r = rec[0]
        assert (r[Fd["AX"]], r[Fd["AY"]]) == tuple(tr["agent"][k]), ctx
        names = ["STEP_COUNT", "STEP_MOVE", "PONE", "PATROL", "UP1", "RIGHT2", "UPD_LONG", "UPD_HORIZ", "RISK",
                 "FIRST_ROOM2", "DIR"]
        for j, nm in enumerate(names):
            assert int(r[Fd[nm]]) == int(tr["scal"][k][j]), ctx + " " + nm
        balls = [(r[Fd["OBX"] + i], r[Fd["OBY"] + i]) for i in range(3)]
        assert balls == list(map(tuple, tr["balls"][k])), ctx
        if r[Fd["O1_VALID"]]:
            assert [(r[Fd["O1X"] + i], r[Fd["O1Y"] + i]) for i in range(3)] == list(map(tuple, tr["o1"][k])), ctx
        else:
            assert (tr["o1"][k] == -1).all(), ctx
        if r[Fd["O2_VALID"]]:
            assert [(r[Fd["O2X"] + i], r[Fd["O2Y"] + i]) for i in range(4)] == list(map(tuple, tr["o2"][k])), ctx
        else:
            assert (tr["o2"][k] == -1).all(), ctx
        assert np.array_equal(_grid_encode(ty[0], co[0]), tr["grid"][k]), ctx
        assert (r[Fd["GOAL_Y"]], r[Fd["GOAL_X"]]) == tuple(tr["pos"][k][2:].astype(int)), ctx
    eng.close()


def test_views_vs_reference_golden(golden_dir):
    """gen_obs_grid(V).encode() for dirs 0-3 x V in {3,5,7,17} (closed-form rotation)."""
    z = np.load(golden_dir + "/views.npz")
    Fd = _fields()
    ncase = len(z["dir"])
    eng = _engine(4, ncase, 17, seed=SEED)
    ty, co, rec = eng.get_state()
    for c in range(ncase):
        g = z["grid"][c]
        ty[c] = np.ascontiguousarray(g[:, :, 0].T).reshape(-1)
        co[c] = np.ascontiguousarray(g[:, :, 1].T).reshape(-1)
        rec[c, Fd["AX"]], rec[c, Fd["AY"]] = z["agent"][c]
        rec[c, Fd["DIR"]] = z["dir"][c]
    eng.set_state(ty, co, rec)
    for V in z["view_sizes"]:
        img = eng.gen_obs(int(V)).cpu().numpy()
        for c in range(ncase):
            assert np.array_equal(img[c], z["img_%03d_V%d" % (c, V)]), (c, int(V))
    eng.close()


_REF_CACHE = {}


def _oracle(variant, N, T, view, env0):
    key = (variant, N, T, view, env0)
    if key not in _REF_CACHE:
        _REF_CACHE.clear()
        _REF_CACHE[key] = orc.rollout(variant, N, T, SEED, env0=env0, view=view)
    return _REF_CACHE[key]


CODE_VALUES = np.array([0.9, -0.9, -0.5, 0.3], np.float32)


def _compare_rollout(variant, N, T, view, env0=0, chunk=None, dense=False, epw=0, supply_actions=False,
                     pipeline=True, codes=False):
    """supply_actions=True feeds the (identical) Philox action stream through HBM, which makes the launch
    eligible for the pipelined kernel (logic wave + emission waves); otherwise the sequential kernel runs."""
    eng = _engine(variant, N, view, seed=SEED, env_id0=env0)
    eng.set_envs_per_wave(epw)
    eng.set_pipeline(pipeline)
    ref = _oracle(variant, N, T, view, env0)
    out = eng.alloc_outputs(T, dense=dense, matrix_codes=codes)
    acts = eng.fill_actions(T) if supply_actions else None
    if chunk is None:
        eng.rollout(T, out, actions=acts)
    else:                                  # same thing in several launches: state must carry over exactly
        for t0 in range(0, T, chunk):
            t1 = min(T, t0 + chunk)
            sub = {k: (v[t0:t1] if v is not None else None) for k, v in out.items()}
            eng.rollout(t1 - t0, sub, actions=None if acts is None else acts[t0:t1])
    torch.cuda.synchronize()
    for k in ("obs", "matrix", "pos", "reward", "terminated", "truncated"):
        got = out[k].cpu().numpy()
        if codes and k == "matrix":                  # uint8 code frames: exact LUT expansion == the float matrix
            assert got.dtype == np.uint8 and int(got.max()) <= 3
            got = CODE_VALUES[got]
        assert got.dtype == ref[k].dtype and got.shape == ref[k].shape, k
        if not np.array_equal(got, ref[k]):
            bad = np.argwhere(got != ref[k])[0]
            raise AssertionError("%s mismatch first at %s: got %s want %s" % (k, bad, got[tuple(bad)], ref[k][tuple(bad)]))
    # final state must equal the oracle-independent sequential kernel's (checked via a second engine)
    eng.close()
    return ref


@pytest.mark.parametrize("variant", [6, 4])
@pytest.mark.parametrize("N,T,view,env0,chunk", [(4096, 200, 17, 0, None), (4096, 300, 17, 0, 128), (1000, 130, 17, 77, 40),
                                                 (17, 64, 17, 5, None), (513, 100, 7, 0, 9), (4099, 40, 9, 1 << 20, None),
                                                 (1500, 70, 17, 9, None), (2048, 130, 17, 0, 64), (300, 90, 17, 4, None)])
def test_pipelined_rollout_vs_oracle(variant, N, T, view, env0, chunk):
    """The pipelined kernel (closed-form logic wave + 15 emission waves per 16 / 8 / 4 / 2 envs, chosen from the batch
    size): bit-exact vs the CPU oracle, incl. ragged N, small views, chunked launches (state hand-over through the
    ping-pong buffers)."""
    _compare_rollout(variant, N, T, view, env0=env0, chunk=chunk, supply_actions=True, pipeline=True)


@pytest.mark.parametrize("variant", [6, 4])
@pytest.mark.parametrize("N,T,view,env0,chunk,dense,pipe", [
    (4096, 200, 17, 0, None, False, True), (1000, 130, 17, 77, 40, False, True), (513, 100, 7, 0, 9, False, True),
    (4096, 200, 17, 0, None, False, False), (777, 90, 17, 3, None, True, False), (65, 64, 3, 7, 16, True, True)])
def test_matrix_code_frames_vs_oracle(variant, N, T, view, env0, chunk, dense, pipe):
    """TW_F_MATRIX_CODE (BASELINE config 5, reduced-precision frames): the uint8 code plane, expanded through its
    4-entry LUT, is bit-identical to the oracle's float matrix -- pipelined and sequential kernels, native
    (304-byte pitch) and dense (289) layouts; every other output unchanged."""
    _compare_rollout(variant, N, T, view, env0=env0, chunk=chunk, dense=dense, supply_actions=pipe, pipeline=pipe,
                     codes=True)


def test_matrix_code_step_api_and_pad():
    eng = _engine(6, 130, 17, seed=SEED)
    ref = _oracle(6, 130, 30, 17, 0)
    out = eng.alloc_outputs(matrix_codes=True)
    raw = padded_rows(out["matrix"], 1, 304)
    assert raw.shape[-1] == 304 and raw.dtype == torch.uint8
    acts = eng.fill_actions(30)
    for t in range(30):
        eng.step(acts[t], out, autoreset=True, policy_idx=True)
        assert np.array_equal(eng.decode_matrix(out["matrix"]).cpu().numpy(), ref["matrix"][t]), t
        assert np.array_equal(out["obs"].cpu().numpy(), ref["obs"][t]), t
    assert int(raw[..., 289:].max()) == 0
    eng.close()


@pytest.mark.parametrize("variant", [6, 4])
def test_pipelined_equals_sequential_state(variant):
    """Same launch through both kernels: identical outputs AND identical final planes / records."""
    N, T = 777, 150
    a, b = _engine(variant, N, 17, seed=SEED), _engine(variant, N, 17, seed=SEED)
    b.set_pipeline(False)
    acts = a.fill_actions(T)
    oa, ob = a.alloc_outputs(T), b.alloc_outputs(T)
    a.rollout(T, oa, actions=acts)
    b.rollout(T, ob, actions=acts)
    torch.cuda.synchronize()
    for k in oa:
        assert torch.equal(oa[k], ob[k]), k
    for x, y, name in zip(_canon(a.get_state()), _canon(b.get_state()), ("type", "colour", "records")):
        assert np.array_equal(x, y), name
    assert a.fallback_count() == 0                 # normal play never leaves the pipelined path


@pytest.mark.parametrize("variant", [6, 4])
@pytest.mark.parametrize("T0", [37, 5])
def test_pipelined_starts_from_a_sequential_kernel_state(variant, T0):
    """Hand-over sequential -> pipelined: the logic wave derives its phases (step_move % 12, patrol bounce phases,
    step_move - step_count, dropped wall blocks, episode coins) from records the SEQUENTIAL kernel wrote mid-episode.
    A: T0 sequential steps, then 100 pipelined; B: everything sequential.  Same outputs, same final state, no fallback."""
    N, T = 1000, 100
    a, b = _engine(variant, N, 17, seed=SEED), _engine(variant, N, 17, seed=SEED)
    b.set_pipeline(False)
    acts = a.fill_actions(T0 + T)
    for eng in (a, b):
        eng.set_pipeline(False)
        eng.rollout(T0, eng.alloc_outputs(T0), actions=acts[:T0].contiguous())
    a.set_pipeline(True)
    oa, ob = a.alloc_outputs(T), b.alloc_outputs(T)
    a.rollout(T, oa, actions=acts[T0:].contiguous())
    b.rollout(T, ob, actions=acts[T0:].contiguous())
    torch.cuda.synchronize()
    for k in oa:
        assert torch.equal(oa[k], ob[k]), k
    for x, y, name in zip(_canon(a.get_state()), _canon(b.get_state()), ("type", "colour", "records")):
        assert np.array_equal(x, y), name
    assert a.fallback_count() == 0


@pytest.mark.parametrize("variant,N,max_steps", [(6, 1, 50), (4, 3, 50), (4, 130, 23), (6, 700, 7)])
def test_pipelined_tiny_batches_and_other_step_caps(variant, N, max_steps):
    """Workgroups with a single real env (N = 1: one env, one padding lane) and MiniGridEnv(max_steps != 50): the
    pipelined kernel against the sequential one, outputs and final state."""
    T = 96
    a = _engine(variant, N, 17, seed=SEED, env_id0=11, max_steps=max_steps)
    b = _engine(variant, N, 17, seed=SEED, env_id0=11, max_steps=max_steps)
    b.set_pipeline(False)
    acts = a.fill_actions(T)
    oa, ob = a.alloc_outputs(T), b.alloc_outputs(T)
    a.rollout(T, oa, actions=acts)
    b.rollout(T, ob, actions=acts)
    torch.cuda.synchronize()
    for k in oa:
        assert torch.equal(oa[k], ob[k]), k
    for x, y, name in zip(_canon(a.get_state()), _canon(b.get_state()), ("type", "colour", "records")):
        assert np.array_equal(x, y), name
    assert a.fallback_count() == 0
    done = (ob["terminated"] | ob["truncated"]) != 0
    assert int(done.sum()) >= (T // max_steps) * N                  # the cap really ends episodes


def test_pipelined_after_a_mid_episode_reset_equals_sequential():
    """MiniGridEnv.reset() in the middle of an episode leaves the Twoarmy flags armed (step_move, pone, ... SURVEY 3.1):
    whatever the pipelined launch makes of such a state (closed form or fallback), it must equal the sequential kernel."""
    N, T = 256, 64
    a, b = _engine(4, N, 17, seed=SEED), _engine(4, N, 17, seed=SEED)
    acts = a.fill_actions(20 + T)
    for eng in (a, b):
        eng.set_pipeline(False)
        eng.rollout(20, eng.alloc_outputs(20), actions=acts[:20].contiguous())
        eng.reset()
    a.set_pipeline(True)
    oa, ob = a.alloc_outputs(T), b.alloc_outputs(T)
    a.rollout(T, oa, actions=acts[20:].contiguous())
    b.rollout(T, ob, actions=acts[20:].contiguous())
    torch.cuda.synchronize()
    sa, sb = _canon(a.get_state()), _canon(b.get_state())
    for x, y, name in zip(sa, sb, ("type", "colour", "records")):
        assert np.array_equal(x, y), name
    # v4 envs reset while their patrols existed hit `cur_pos is None` (TypeError, twoarmy_v4.py:122-124): raised steps
    # leave their output rows unwritten, every other env's rows must agree
    ok = torch.from_numpy(sa[2][:, _fields()["ERROR"]] == 0).cuda()
    assert 0 < int(ok.sum()) < N
    for k in oa:
        assert torch.equal(oa[k][:, ok], ob[k][:, ok]), k


def _canon(state):
    """Zero record fields that are meaningless in the current state (cur_pos of unspawned patrols, wall offsets
    before the drop): the reference objects simply do not exist then, the two kernels keep different leftovers."""
    Fd = _fields()
    ty, co, rec = state
    rec = rec.copy()
    rec[rec[:, Fd["O1_VALID"]] == 0, Fd["O1X"]:Fd["O1X"] + 6] = 0
    rec[rec[:, Fd["O2_VALID"]] == 0, Fd["O2X"]:Fd["O2X"] + 8] = 0
    rec[rec[:, Fd["PONE"]] == 0, Fd["WALL_I1"]:Fd["WALL_I1"] + 2] = 0
    return ty, co, rec


def test_pipelined_falls_back_on_abnormal_state_and_illegal_actions():
    """Anything outside normal play must come out exactly as the sequential kernel computes it:
    (a) injected drifted balls, (b) env actions 4/5 (AttributeError in the reference), (c) a foreign grid cell."""
    Fd = _fields()
    N, T = 96, 40
    for case in ("drift", "illegal", "grid"):
        a, b = _engine(6, N, 17, seed=SEED), _engine(6, N, 17, seed=SEED)
        b.set_pipeline(False)
        acts = a.fill_actions(T)
        policy_idx = True
        if case == "illegal":
            acts = acts.clone(); acts[3, 5] = 4; acts[7, 50] = 5; policy_idx = False
        for eng in (a, b):
            ty, co, rec = eng.get_state()
            if case == "drift":
                for n in (0, 17, 95):
                    ty[n, 8 * 17 + 7] = 1; co[n, 8 * 17 + 7] = 0; ty[n, 8 * 17 + 10] = 6; co[n, 8 * 17 + 10] = 4
                    rec[n, Fd["OBX"]:Fd["OBX"] + 3] = [8, 9, 10]
            if case == "grid":
                ty[40, 3 * 17 + 3] = 2; co[40, 3 * 17 + 3] = 5
            eng.set_state(ty, co, rec)
        oa, ob = a.alloc_outputs(T), b.alloc_outputs(T)
        a.rollout(T, oa, actions=acts, policy_idx=policy_idx)
        b.rollout(T, ob, actions=acts, policy_idx=policy_idx)
        torch.cuda.synchronize()
        sa, sb = _canon(a.get_state()), _canon(b.get_state())
        for x, y, name in zip(sa, sb, ("type", "colour", "records")):
            assert np.array_equal(x, y), (case, name)
        if case != "illegal":                      # raised steps leave their output rows unwritten
            for k in oa:
                assert torch.equal(oa[k], ob[k]), (case, k)
        else:
            assert int(sa[2][:, Fd["ERROR"]].max()) == 1
        assert a.fallback_count() == 1 and b.fallback_count() == 0, (case, a.fallback_count())   # one launch, re-run once


@pytest.mark.parametrize("variant", [6, 4])
@pytest.mark.parametrize("dense,epw", [(False, 0), (False, 1), (False, 4), (True, 0), (True, 1)])
def test_rollout_4096_vs_oracle(variant, dense, epw):
    """BASELINE config 2: 4096 envs, fused T-step launch, bit-exact obs/matrix/reward/done vs the CPU oracle.
    Native (16-byte pitched, register-packed path) and dense (generic path) layouts; 1/2/4 envs per wave."""
    ref = _compare_rollout(variant, 4096, 200, 17, dense=dense, epw=epw)
    assert ref["truncated"].sum() > 0


def test_native_layout_pad_is_zero():
    eng = _engine(4, 130, 17, seed=SEED)
    out = eng.alloc_outputs(40)
    raw_o, raw_m = padded_rows(out["obs"], 2, 880), padded_rows(out["matrix"], 2, 292)
    raw_o.fill_(0xAB)
    raw_m.fill_(7.0)
    eng.rollout(40, out)
    torch.cuda.synchronize()
    assert raw_o.shape[-1] == 880 and raw_m.shape[-1] == 292
    assert int(raw_o[..., 867:].max()) == 0
    assert float(raw_m[..., 289:].abs().max()) == 0.0


@pytest.mark.parametrize("variant,view,N,T,env0", [(6, 7, 513, 130, 0), (4, 7, 257, 150, 12345), (4, 3, 65, 64, 7),
                                                   (6, 5, 1, 120, 4095), (4, 17, 1000, 100, 1 << 20),
                                                   (4, 15, 333, 70, 5), (6, 9, 4099, 66, 0)])
@pytest.mark.parametrize("dense,epw", [(False, 0), (True, 2), (False, 4)])
def test_rollout_shapes_vs_oracle(variant, view, N, T, env0, dense, epw):
    """ragged N (not a multiple of 2/4/64), small views, sharded env-id offsets, both layouts"""
    _compare_rollout(variant, N, T, view, env0=env0, dense=dense, epw=epw)


@pytest.mark.parametrize("variant", [6, 4])
def test_rollout_chunked_equals_single(variant):
    _compare_rollout(variant, 300, 96, 17, chunk=1)      # tw_rollout(T=1) x 96 == oracle
    _compare_rollout(variant, 300, 100, 17, chunk=33)
    _compare_rollout(variant, 301, 100, 17, chunk=33, dense=True, epw=4)


def test_step_api_equals_rollout():
    """tw_step with explicit actions == tw_rollout with Philox actions (fill_actions gives the same stream)."""
    N, T = 777, 60
    a = _engine(4, N, 17, seed=SEED)
    b = _engine(4, N, 17, seed=SEED)
    acts = a.fill_actions(T)
    out_a = a.alloc_outputs(T)
    a.rollout(T, out_a)
    out_b = b.alloc_outputs(T)
    for t in range(T):
        sub = {k: v[t] for k, v in out_b.items()}
        b.step(acts[t], sub, autoreset=True, policy_idx=True)
    torch.cuda.synchronize()
    for k in out_a:
        assert torch.equal(out_a[k], out_b[k]), k
    sa, sb = a.get_state(), b.get_state()
    for x, y in zip(sa, sb):
        assert np.array_equal(x, y)


def test_full_size_properties():
    """Size-independent properties at the benchmark size (4096 envs x 512 steps, v4):
    rewards in the 5-value set; done => step_count reset; obs window consistent with the state matrix."""
    N, T = 4096, 512
    eng = _engine(4, N, 17, seed=SEED)
    out = eng.alloc_outputs(T)
    eng.rollout(T, out)
    torch.cuda.synchronize()
    r = out["reward"]
    vals = torch.tensor([-0.01, -0.1, -0.9, 0.2, 0.9], device="cuda")
    assert bool((r.unsqueeze(-1) == vals).any(-1).all())
    m = out["matrix"]
    assert bool(((m == 0.9) | (m == -0.9) | (m == -0.5) | (m == 0.3)).all())
    assert bool(((m == 0.3).sum(-1) == 1).all())                       # exactly one agent cell
    pos = out["pos"].long()
    agent_idx = pos[..., 0] * 17 + pos[..., 1]
    assert bool((m.gather(-1, agent_idx.unsqueeze(-1)).squeeze(-1) == 0.3).all())
    # agent's own view cell is always empty (1,0,0); dir == 3 -> view cell (8,16)
    assert bool((out["obs"][:, :, 8, 16, 0] == 1).all())
    # done iff (terminated | truncated); episodes never exceed 50 steps
    done = (out["terminated"] | out["truncated"]).bool()
    run = torch.zeros(N, dtype=torch.long, device="cuda")
    mx = 0
    for t in range(T):
        run += 1
        mx = max(mx, int(run.max()))
        run[done[t]] = 0
    assert mx <= 50
    _, _, rec = eng.get_state()
    assert int(rec[:, _fields()["ERROR"]].max()) == 0
    assert int(rec[:, _fields()["T"]].min()) == T


def test_slab_free_then_realloc_writes_every_row():
    """Regression for the slab incident ("engine slab"): allocate a slab, roll out into it, drop it (so
    tw_free_outputs unmaps / releases its chunks), allocate again, prefill the new slab with 0x77 and run the pipelined
    and the sequential rollout into fresh slabs: equal outputs, equal to the oracle, no row left at the prefill value.
    Runs ONCE; whatever tw_free_outputs does with the address range (kept reserved today) must keep this green."""
    import gc
    N, T = 4096, 128
    ref = _oracle(6, N, T, 17, 0)
    eng = _engine(6, N, 17, seed=SEED)
    acts = eng.fill_actions(T)
    state0 = eng.get_state()
    out = eng.alloc_outputs(T)
    assert out["matrix"]._tw_layout.startswith("2048-byte records")
    first_ptr = out["matrix"].data_ptr()
    eng.rollout(T, out, actions=acts)
    torch.cuda.synchronize()
    del out
    gc.collect()                                           # last tensor gone -> _OutputSlab.__del__ -> tw_free_outputs
    results = {}
    for pipe in (True, False):
        eng.set_state(*state0)
        eng.set_pipeline(pipe)
        out = eng.alloc_outputs(T)
        rows = padded_rows(out["matrix"], 2, 292)          # the whole 2048-byte record starts at the matrix row
        rec = rows.view(torch.uint8).as_strided((T, N, 2048), (N * 2048, 2048, 1))
        rec.fill_(0x77)
        for k in ("pos", "reward", "terminated", "truncated"):
            out[k].view(torch.uint8).fill_(0x77)
        torch.cuda.synchronize()
        eng.rollout(T, out, actions=acts)
        torch.cuda.synchronize()
        # a row the kernel never reached still reads 0x77 in its first matrix word (a float matrix cell is one of
        # +-0.9, -0.5, 0.3: never 0x77777777)
        first_words = rec[..., :4].contiguous().view(torch.int32)
        assert int((first_words == 0x77777777).sum()) == 0, "pipe=%s: rows never written after a slab re-allocation" % pipe
        for k in ("obs", "matrix", "pos", "reward", "terminated", "truncated"):
            assert np.array_equal(out[k].cpu().numpy(), ref[k]), (k, pipe)
        results[pipe] = {k: out[k].clone() for k in out}
        ptr = out["matrix"].data_ptr()
        del out, rows, rec, first_words
        gc.collect()
    for k in results[True]:
        assert torch.equal(results[True][k], results[False][k]), k
    assert eng.fallback_count() == 0
    eng.close()
    assert first_ptr and ptr
