"""TwoarmyEngine: torch-tensor front end of the HIP engine handle (include/twoarmy.h).

PyTorch is plumbing here (device memory + streams); all env compute is in libtwoarmy_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import FIELDS, TW_CELLS, TW_DRAW_WORDS, TW_F_AUTORESET, TW_F_POLICY_IDX, TW_REC_WORDS

REWARD_VALUES = (-0.01, -0.1, -0.9, 0.2, 0.9)


def _ptr(t, dtype=None):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "engine buffers must be contiguous device tensors"
    if dtype is not None:
        assert t.dtype == dtype, "expected %s, got %s" % (dtype, t.dtype)
    return C.c_void_p(t.data_ptr())


class TwoarmyEngine:
    """N independent MiniGrid-Twoarmy envs living in HBM on one GPU.

    variant: "v4"/4 (hard) or "v6"/6 (easy) -- gym ids MiniGrid-twoarmy-17x17-v{4,6}
    (reference gym_minigrid/__init__.py:6-21).
    """

    def __init__(self, variant, num_envs, view_size=17, device=None, seed=9981, env_id0=0):
        variant = {"v4": 4, "v6": 6}.get(variant, variant)
        if not torch.cuda.is_available():
            raise _lib.TwoarmyLibraryError("TwoarmyEngine needs a GPU (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.variant, self.num_envs, self.view_size = variant, int(num_envs), int(view_size)
        self.seed, self.env_id0 = int(seed), int(env_id0)
        self._h = C.c_void_p()
        lib = _lib.lib()
        with torch.cuda.device(self.device):
            torch.cuda.current_stream().synchronize()
            _lib.check(lib.tw_create(C.byref(self._h), variant, self.num_envs, self.view_size,
                                     self.device.index or 0, self.seed, self.env_id0), "tw_create")

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().tw_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ buffers
    def alloc_outputs(self, T=None, obs=True, matrix=True):
        N, V = self.num_envs, self.view_size
        lead = (N,) if T is None else (T, N)
        d = self.device
        return dict(
            obs=torch.empty(lead + (V, V, 3), dtype=torch.uint8, device=d) if obs else None,
            matrix=torch.empty(lead + (TW_CELLS,), dtype=torch.float32, device=d) if matrix else None,
            pos=torch.empty(lead + (2,), dtype=torch.float32, device=d),
            reward=torch.empty(lead, dtype=torch.float32, device=d),
            terminated=torch.empty(lead, dtype=torch.uint8, device=d),
            truncated=torch.empty(lead, dtype=torch.uint8, device=d),
        )

    # ------------------------------------------------------------------ ops
    def reset(self, mask=None, obs=None):
        _lib.check(_lib.lib().tw_reset(self._h, _ptr(mask, torch.uint8), _ptr(obs, torch.uint8), self._stream()),
                   "tw_reset")
        return obs

    def step(self, actions, out, draws=None, autoreset=False, policy_idx=False):
        flags = (TW_F_AUTORESET if autoreset else 0) | (TW_F_POLICY_IDX if policy_idx else 0)
        assert actions.shape == (self.num_envs,)
        if draws is not None:
            assert draws.shape == (self.num_envs, TW_DRAW_WORDS) and draws.dtype == torch.int32
        _lib.check(_lib.lib().tw_step(
            self._h, _ptr(actions, torch.int32), _ptr(draws), _ptr(out.get("obs"), torch.uint8),
            _ptr(out.get("matrix"), torch.float32), _ptr(out.get("pos"), torch.float32),
            _ptr(out.get("reward"), torch.float32), _ptr(out.get("terminated"), torch.uint8),
            _ptr(out.get("truncated"), torch.uint8), flags, self._stream()), "tw_step")
        return out

    def rollout(self, T, out, actions=None, draws=None, autoreset=True, policy_idx=True):
        flags = (TW_F_AUTORESET if autoreset else 0) | (TW_F_POLICY_IDX if policy_idx else 0)
        if actions is not None:
            assert actions.shape == (T, self.num_envs)
        if draws is not None:
            assert draws.shape == (T, self.num_envs, TW_DRAW_WORDS) and draws.dtype == torch.int32
        for k in ("obs", "matrix", "pos", "reward", "terminated", "truncated"):
            t = out.get(k)
            assert t is None or t.shape[:2] == (T, self.num_envs), k
        _lib.check(_lib.lib().tw_rollout(
            self._h, T, _ptr(actions, torch.int32), _ptr(draws), _ptr(out.get("obs"), torch.uint8),
            _ptr(out.get("matrix"), torch.float32), _ptr(out.get("pos"), torch.float32),
            _ptr(out.get("reward"), torch.float32), _ptr(out.get("terminated"), torch.uint8),
            _ptr(out.get("truncated"), torch.uint8), flags, self._stream()), "tw_rollout")
        return out

    def fill_actions(self, T):
        a = torch.empty((T, self.num_envs), dtype=torch.int32, device=self.device)
        _lib.check(_lib.lib().tw_fill_actions(self._h, T, _ptr(a), self._stream()), "tw_fill_actions")
        return a

    def gen_obs(self, view_size=None):
        V = view_size or self.view_size
        obs = torch.empty((self.num_envs, V, V, 3), dtype=torch.uint8, device=self.device)
        _lib.check(_lib.lib().tw_gen_obs(self._h, V, _ptr(obs), self._stream()), "tw_gen_obs")
        return obs

    def time_rollout(self, T, out, actions=None, autoreset=True, iters=10):
        """Mean kernel time (ms) of one tw_rollout launch, HIP events on the current stream."""
        ms = C.c_float()
        flags = (TW_F_AUTORESET if autoreset else 0) | TW_F_POLICY_IDX
        _lib.check(_lib.lib().tw_time_rollout(
            self._h, T, _ptr(actions, torch.int32), _ptr(out.get("obs")), _ptr(out.get("matrix")),
            _ptr(out.get("pos")), _ptr(out.get("reward")), _ptr(out.get("terminated")), _ptr(out.get("truncated")),
            flags, iters, self._stream(), C.byref(ms)), "tw_time_rollout")
        return ms.value

    # ------------------------------------------------------------------ state
    def get_state(self):
        """(type uint8[N,289], colour uint8[N,289], records int32[N,48]) as numpy (host copies)."""
        N = self.num_envs
        ty = np.empty((N, TW_CELLS), np.uint8)
        co = np.empty((N, TW_CELLS), np.uint8)
        rec = np.empty((N, TW_REC_WORDS), np.int32)
        _lib.check(_lib.lib().tw_get_state_host(self._h, ty.ctypes.data_as(C.c_void_p), co.ctypes.data_as(C.c_void_p),
                                                rec.ctypes.data_as(C.c_void_p)), "tw_get_state_host")
        return ty, co, rec

    def set_state(self, type_plane=None, colour_plane=None, records=None):
        def p(a, dt, shape):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt)
            assert a.shape == shape
            keep.append(a)
            return a.ctypes.data_as(C.c_void_p)
        keep = []
        N = self.num_envs
        _lib.check(_lib.lib().tw_set_state_host(self._h, p(type_plane, np.uint8, (N, TW_CELLS)),
                                                p(colour_plane, np.uint8, (N, TW_CELLS)),
                                                p(records, np.int32, (N, TW_REC_WORDS))), "tw_set_state_host")

    def state_tensors(self):
        """Zero-copy device views of the engine's SoA state (type, colour, records)."""
        raise NotImplementedError("use get_state()/set_state(); device views arrive with the DLPack bridge")

    @staticmethod
    def field(records, name, k=0):
        return records[..., FIELDS[name] + k]
