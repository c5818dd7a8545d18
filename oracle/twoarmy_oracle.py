"""ctypes binding of oracle/twoarmy_oracle.c.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtwoarmy_oracle.so")
NC = 289


class TwEnv(C.Structure):
    _fields_ = [
        ("type", C.c_uint8 * NC), ("colour", C.c_uint8 * NC), ("state", C.c_uint8 * NC),
        ("ax", C.c_int32), ("ay", C.c_int32), ("dir", C.c_int32),
        ("step_count", C.c_int32), ("max_steps", C.c_int32), ("variant", C.c_int32),
        ("step_move", C.c_int32), ("pone", C.c_int32), ("patrol", C.c_int32), ("up1", C.c_int32),
        ("right2", C.c_int32), ("upd_long", C.c_int32), ("upd_horiz", C.c_int32),
        ("risk_count", C.c_int32), ("first_to_room2", C.c_int32),
        ("ob_x", C.c_int32 * 3), ("ob_y", C.c_int32 * 3),
        ("o1_x", C.c_int32 * 3), ("o1_y", C.c_int32 * 3), ("o1_valid", C.c_int32),
        ("o2_x", C.c_int32 * 4), ("o2_y", C.c_int32 * 4), ("o2_valid", C.c_int32),
        ("goal_x", C.c_int32), ("goal_y", C.c_int32),
        ("t", C.c_uint32), ("error", C.c_int32),
        ("reward_code", C.c_int32), ("terminated", C.c_int32), ("truncated", C.c_int32),
    ]


REWARD_VALUE = (-0.01, -0.1, -0.9, 0.2, 0.9)
ERROR_NAME = {0: None, 1: "AttributeError", 2: "AssertionError", 3: "TypeError"}


def build(force=False):
    src = os.path.join(_HERE, "twoarmy_oracle.c")
    if force or not os.path.exists(_SO) or (os.path.exists(src) and os.path.getmtime(_SO) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        assert _lib.tw_oracle_sizeof_env() == C.sizeof(TwEnv), "ctypes struct out of sync with C"
        _lib.tw_oracle_draw_word.restype = C.c_uint32
        _lib.tw_oracle_draw_word.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        _lib.tw_oracle_step.restype = C.c_int
        _lib.tw_oracle_step.argtypes = [C.POINTER(TwEnv), C.c_int, C.c_void_p, C.c_uint64, C.c_uint32,
                                        C.c_int, C.c_void_p]
        _lib.tw_oracle_gen_obs.argtypes = [C.POINTER(TwEnv), C.c_int, C.c_void_p]
        _lib.tw_oracle_matrix.argtypes = [C.POINTER(TwEnv), C.c_void_p]
        _lib.tw_oracle_rollout.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int,
                                           C.c_void_p, C.c_void_p] + [C.c_void_p] * 6 + [C.c_int]
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleEnv:
    """Single reference-semantics Twoarmy env (no auto-reset), mirroring the reference attributes."""

    def __init__(self, variant=6, seed=0, env_id=0, view=17):
        self.e = TwEnv()
        self.seed, self.env_id, self.view = seed, env_id, view
        lib().tw_oracle_init(C.byref(self.e), variant)

    def reset(self):
        lib().tw_oracle_reset(C.byref(self.e))
        return self.gen_obs()

    def gen_obs(self, view=None, direction=None):
        V = view or self.view
        out = np.empty((V, V, 3), np.uint8)
        if direction is not None:
            saved, self.e.dir = self.e.dir, direction
        lib().tw_oracle_gen_obs(C.byref(self.e), V, _ptr(out))
        if direction is not None:
            self.e.dir = saved
        return out

    def step(self, action, draws=None):
        """Returns (obs, reward, terminated, truncated, error_name)."""
        obs = np.empty((self.view, self.view, 3), np.uint8)
        d = None if draws is None else np.ascontiguousarray(draws, dtype=np.uint32)
        err = lib().tw_oracle_step(C.byref(self.e), int(action), _ptr(d), self.seed, self.env_id,
                                   self.view, _ptr(obs))
        if err:
            return None, None, None, None, ERROR_NAME[err]
        return obs, REWARD_VALUE[self.e.reward_code], bool(self.e.terminated), bool(self.e.truncated), None

    def matrix(self):
        m = np.empty(NC, np.float32)
        lib().tw_oracle_matrix(C.byref(self.e), _ptr(m))
        return m

    def grid_encode(self):
        """Grid.encode() of the full grid: uint8[17,17,3] indexed [x][y][c]."""
        t = np.frombuffer(self.e.type, np.uint8).reshape(17, 17)
        c = np.frombuffer(self.e.colour, np.uint8).reshape(17, 17)
        s = np.frombuffer(self.e.state, np.uint8).reshape(17, 17)
        return np.stack([t.T, c.T, s.T], axis=-1).copy()

    def flags(self):
        e = self.e
        return dict(agent_pos=(e.ax, e.ay), step_count=e.step_count, step_move=e.step_move, pone=bool(e.pone),
                    patrol=bool(e.patrol), up1=bool(e.up1), right2=bool(e.right2),
                    Update_longitudinal=bool(e.upd_long), Update_horizontal=bool(e.upd_horiz),
                    risk_count=e.risk_count, first_to_room2=bool(e.first_to_room2),
                    ball_x=tuple(e.ob_x), ball_y=tuple(e.ob_y))


def timed_rollout(variant, N, seconds, seed, view=17, chunk=64, threads=1):
    """CPU baseline leg of bench.py: keep stepping the same N envs (state carried across chunks, outputs
    written to reused [chunk,n,...] buffers) for about `seconds`; returns (env_steps, elapsed_seconds).
    threads > 1 shards the envs over that many host threads (ctypes releases the GIL during the C call),
    each with its own env range [env0, env0+n) and its own output buffers."""
    import threading
    import time
    lib()
    threads = max(1, min(int(threads), N))
    bounds = [N * k // threads for k in range(threads + 1)]
    counts = [0] * threads
    start = threading.Barrier(threads)
    t0 = [0.0]

    def work(k):
        e0, n = bounds[k], bounds[k + 1] - bounds[k]
        envs = (TwEnv * n)()
        for i in range(n):
            lib().tw_oracle_init(C.byref(envs[i]), variant)
        obs = np.empty((chunk, n, view, view, 3), np.uint8)
        mat = np.empty((chunk, n, NC), np.float32)
        pos = np.empty((chunk, n, 2), np.float32)
        rew = np.empty((chunk, n), np.float32)
        te = np.empty((chunk, n), np.uint8)
        tr = np.empty((chunk, n), np.uint8)
        def go():
            lib().tw_oracle_rollout(variant, n, chunk, seed, e0, view, None, C.cast(envs, C.c_void_p), _ptr(obs),
                                    _ptr(mat), _ptr(pos), _ptr(rew), _ptr(te), _ptr(tr), 1)
        go()                                   # untimed: first touch of the output pages
        if start.wait() == 0:
            t0[0] = time.perf_counter()
        start.wait()
        while True:
            go()
            counts[k] += n * chunk
            if time.perf_counter() - t0[0] >= seconds:
                return

    if threads == 1:
        work(0)
    else:
        ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    return sum(counts), time.perf_counter() - t0[0]


def rollout(variant, N, T, seed, env0=0, view=17, actions=None, autoreset=True,
            want_obs=True, want_matrix=True):
    """Batched CPU rollout; returns dict of [T,N,...] arrays."""
    out = {
        "obs": np.empty((T, N, view, view, 3), np.uint8) if want_obs else None,
        "matrix": np.empty((T, N, NC), np.float32) if want_matrix else None,
        "pos": np.empty((T, N, 2), np.float32),
        "reward": np.empty((T, N), np.float32),
        "terminated": np.empty((T, N), np.uint8),
        "truncated": np.empty((T, N), np.uint8),
    }
    a = None if actions is None else np.ascontiguousarray(actions, dtype=np.int32)
    lib().tw_oracle_rollout(variant, N, T, seed, env0, view, _ptr(a), None,
                            _ptr(out["obs"]), _ptr(out["matrix"]), _ptr(out["pos"]), _ptr(out["reward"]),
                            _ptr(out["terminated"]), _ptr(out["truncated"]), int(autoreset))
    return out
