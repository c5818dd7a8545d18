"""TwoarmyEngine: torch-tensor front end of the HIP engine handle (include/twoarmy.h).

PyTorch is plumbing here (device memory + streams); all env compute is in libtwoarmy_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import (FIELDS, TW_CELLS, TW_DRAW_WORDS, TW_F_AUTORESET, TW_F_MATRIX_CODE, TW_F_POLICY_IDX, TW_F_SLAB_HIPMALLOC,
                   TW_REC_WORDS)

REWARD_VALUES = (-0.01, -0.1, -0.9, 0.2, 0.9)
MAT_PITCH = 292                       # floats per env matrix in the native layout (289 + 3 zero pad)
MATC_PITCH = 304                      # bytes per env matrix in the code layout (TW_F_MATRIX_CODE; 289 + 15 pad)
MATRIX_CODE_VALUES = (0.9, -0.9, -0.5, 0.3)     # code -> Env_transact.matrix_env value (free/goal, wall, ball, agent)


def obs_pitch_for(view):
    """Bytes per env image in the native layout: V*V*3 rounded up to 16 (880 for V=17)."""
    return (view * view * 3 + 15) // 16 * 16


def _ptr(t, dtype=None):
    if t is None:
        return None
    assert t.is_cuda, "engine buffers must be device tensors"
    if dtype is not None:
        assert t.dtype == dtype, "expected %s, got %s" % (dtype, t.dtype)
    return C.c_void_p(t.data_ptr())


def _dense(t, dtype):
    assert t is None or (t.is_contiguous() and t.dtype == dtype), "expected contiguous %s" % dtype
    return _ptr(t)


def _pitched(t, lead, inner_shape, dtype):
    """(pointer, pitch in elements) of a [*lead, *inner_shape] tensor whose rows may be padded."""
    if t is None:
        return None, 0
    assert t.is_cuda and t.dtype == dtype, "expected %s device tensor" % dtype
    nlead = len(lead)
    assert tuple(t.shape) == tuple(lead) + tuple(inner_shape), (tuple(t.shape), lead, inner_shape)
    inner = 1
    for d in range(t.dim() - 1, nlead - 1, -1):            # inner dims must be dense
        assert t.stride(d) == inner or t.shape[d] == 1, "inner dims of an engine buffer must be dense"
        inner *= t.shape[d]
    # row pitch from the innermost leading dim of size > 1 (strides of size-1 dims are arbitrary)
    pitch, mult = None, 1
    for d in range(nlead - 1, -1, -1):
        if t.shape[d] > 1:
            if pitch is None:
                assert t.stride(d) % mult == 0
                pitch = t.stride(d) // mult
            else:
                assert t.stride(d) == pitch * mult, "leading dims must be dense"
        mult *= t.shape[d]
    if pitch is None:
        pitch = inner
    assert pitch >= inner
    return C.c_void_p(t.data_ptr()), int(pitch)


def padded_rows(t, nlead, width):
    """The padded rows behind a native-layout output: [*lead, width] view of every row including its pad bytes /
    floats (t is the [*lead, ...] strided view handed out by alloc_outputs; nlead = len(lead); width = 880 / 292 / 304)."""
    lead = tuple(t.shape[:nlead])
    assert width <= t.stride(nlead - 1)
    strides = tuple(t.stride(d) for d in range(nlead)) + (1,)
    return t.as_strided(lead + (width,), strides)


class _DevSpan:
    """A span of a library-owned device slab, exported through __cuda_array_interface__ so that torch wraps it without
    copying; torch keeps this object (and through it the slab) alive as long as any view of the tensor lives."""

    def __init__(self, owner, ptr, nbytes):
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class _OutputSlab:
    """One tw_alloc_outputs slab; freed (tw_free_outputs) when the last tensor carved from it is gone."""

    def __init__(self, engine, T, flags):
        self.out = _lib.TwOutputs()
        self.device = engine.device
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().tw_alloc_outputs(engine._h, int(T), int(flags), C.byref(self.out)), "tw_alloc_outputs")
        self.backing = int(self.out.backing)

    def tensor(self, ptr, nbytes, dtype):
        t = torch.as_tensor(_DevSpan(self, ptr, nbytes), device=self.device)
        if t.data_ptr() != int(ptr) or t.device != self.device:
            raise _lib.TwoarmyLibraryError("torch did not alias the engine slab (got %s at %#x for %#x on %s)"
                                           % (t.device, t.data_ptr(), int(ptr), self.device))
        return t.view(dtype)

    def __del__(self):
        try:
            if self.out.slab:
                rc = _lib.lib().tw_free_outputs(C.byref(self.out))
                if rc != 0:
                    import sys
                    print("tw_free_outputs failed: rc=%d %s" % (rc, _lib.lib().tw_last_error_message().decode()), file=sys.stderr)
        except Exception:
            pass


class TwoarmyEngine:
    """N independent MiniGrid-Twoarmy envs living in HBM on one GPU.

    variant: "v4"/4 (hard) or "v6"/6 (easy) -- gym ids MiniGrid-twoarmy-17x17-v{4,6}
    (reference gym_minigrid/__init__.py:6-21).
    """

    def __init__(self, variant, num_envs, view_size=17, device=None, seed=9981, env_id0=0, max_steps=50):
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
        self.max_steps = int(max_steps)
        if self.max_steps != 50:              # MiniGridEnv(max_steps=...) (minigrid.py:866-945; Twoarmy passes 50)
            if self.max_steps <= 0:
                raise ValueError("max_steps must be positive")
            _, _, rec = self.get_state()
            rec[:, FIELDS["MAX_STEPS"]] = self.max_steps
            self.set_state(records=rec)

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

    def set_pipeline(self, enable):
        _lib.check(_lib.lib().tw_set_pipeline(self._h, int(bool(enable))), "tw_set_pipeline")

    def fallback_count(self):
        """Pipelined launches re-run by the sequential fallback so far (0 in normal play)."""
        n = C.c_int()
        _lib.check(_lib.lib().tw_fallback_count(self._h, C.byref(n)), "tw_fallback_count")
        return n.value

    def set_envs_per_wave(self, e):
        _lib.check(_lib.lib().tw_set_envs_per_wave(self._h, int(e)), "tw_set_envs_per_wave")

    # ------------------------------------------------------------------ buffers
    def alloc_outputs(self, T=None, obs=True, matrix=True, dense=False, matrix_codes=False, slab=None):
        """Output tensors for step (T=None -> [N,...]) or rollout ([T,N,...]).

        Native layout (dense=False): obs rows padded to 16 bytes (880 for V=17) and matrix rows to
        292 floats; the returned tensors are [..., V, V, 3] / [..., 289] strided views of them.
        matrix_codes=True: the matrix is uint8 codes (rows of 304 bytes natively), see TW_F_MATRIX_CODE;
        a uint8 matrix tensor selects that mode in step/rollout.
        slab (default: on for the native layout with every output): the buffers are carved out of ONE slab the
        library allocates (tw_alloc_outputs) -- the same placement for every user of the C ABI; the tensors keep the
        slab alive.  slab=False takes them from torch's caching allocator."""
        N, V = self.num_envs, self.view_size
        lead = (N,) if T is None else (T, N)
        d = self.device
        nb = V * V * 3
        if slab is None:
            slab = (not dense) and obs and matrix
        if slab:
            assert not dense and obs and matrix, "the engine slab holds the native layout with every output"
            return self._slab_outputs(T, lead, matrix_codes)
        if dense:
            o = torch.empty(lead + (V, V, 3), dtype=torch.uint8, device=d) if obs else None
            m = torch.empty(lead + (TW_CELLS,), dtype=torch.uint8 if matrix_codes else torch.float32, device=d) \
                if matrix else None
        else:
            o = torch.empty(lead + (obs_pitch_for(V),), dtype=torch.uint8, device=d)[..., :nb].view(lead + (V, V, 3)) \
                if obs else None
            if not matrix:
                m = None
            elif matrix_codes:
                m = torch.empty(lead + (MATC_PITCH,), dtype=torch.uint8, device=d)[..., :TW_CELLS]
            else:
                m = torch.empty(lead + (MAT_PITCH,), dtype=torch.float32, device=d)[..., :TW_CELLS]
        return dict(
            obs=o, matrix=m,
            pos=torch.empty(lead + (2,), dtype=torch.float32, device=d),
            reward=torch.empty(lead, dtype=torch.float32, device=d),
            terminated=torch.empty(lead, dtype=torch.uint8, device=d),
            truncated=torch.empty(lead, dtype=torch.uint8, device=d),
        )

    def _slab_outputs(self, T, lead, matrix_codes):
        N, V = self.num_envs, self.view_size
        flags = TW_F_MATRIX_CODE if matrix_codes else 0
        owner = _OutputSlab(self, 1 if T is None else T, flags)
        fell_back = False
        try:
            owner.tensor(min(owner.out.obs, owner.out.matrix), 16, torch.uint8)
        except _lib.TwoarmyLibraryError as exc:
            # a mapped slab torch cannot wrap in place (seen nowhere so far; multi-GPU ranks are the untested case):
            # the same layout from plain hipMalloc memory, whose pointer attributes torch certainly understands.
            # Never silent: a warning here, `_tw_fallback` on the tensors, `config.slab_backing` in bench.py's line
            # (ranks that disagree abort the run), TW_SLAB_STRICT=1 turns it into an error.
            import os
            import warnings
            if os.environ.get("TW_SLAB_STRICT", "0") == "1":
                raise
            warnings.warn("engine slab: falling back to hipMalloc backing (%s)" % exc, RuntimeWarning)
            fell_back = True
            owner = _OutputSlab(self, 1 if T is None else T, flags | TW_F_SLAB_HIPMALLOC)
        o = owner.out
        TN = (1 if T is None else T) * N
        nb = V * V * 3
        # one stream of records (include/twoarmy.h): float frames = matrix row | image row, code frames = image row | code row
        rb = int(o.obs_pitch)
        rec = owner.tensor(min(o.obs, o.matrix), TN * rb, torch.uint8).view(lead + (rb,))
        orow = obs_pitch_for(V)
        if matrix_codes:
            assert o.matrix == o.obs + orow and o.mat_pitch == rb == orow + MATC_PITCH
            obs = rec[..., :nb].view(lead + (V, V, 3))
            m = rec[..., orow:orow + TW_CELLS]
        else:
            assert o.obs == o.matrix + MAT_PITCH * 4 and o.mat_pitch * 4 == rb == MAT_PITCH * 4 + orow
            m = rec[..., :MAT_PITCH * 4].view(torch.float32)[..., :TW_CELLS]
            obs = rec[..., MAT_PITCH * 4:MAT_PITCH * 4 + nb].view(lead + (V, V, 3))
        m._tw_layout = ("%d-byte records (image | codes)" % rb if matrix_codes else "%d-byte records (matrix | image)" % rb) + \
            (", hipMalloc" if o.backing == 0 else ", 2 MiB mapped chunks")
        m._tw_fallback = fell_back
        return dict(obs=obs, matrix=m,
                    pos=owner.tensor(o.pos, TN * 8, torch.float32).view(lead + (2,)),
                    reward=owner.tensor(o.reward, TN * 4, torch.float32).view(lead),
                    terminated=owner.tensor(o.terminated, TN, torch.uint8).view(lead),
                    truncated=owner.tensor(o.truncated, TN, torch.uint8).view(lead))

    # ------------------------------------------------------------------ ops
    def reset(self, mask=None, obs=None):
        V = self.view_size
        op, opitch = _pitched(obs, (self.num_envs,), (V, V, 3), torch.uint8)
        _lib.check(_lib.lib().tw_reset(self._h, _dense(mask, torch.uint8), op, opitch, self._stream()), "tw_reset")
        return obs

    @staticmethod
    def _matrix_arg(m, lead, flags):
        """(pointer, pitch, flags): a uint8 matrix tensor means code frames (pitch in bytes)."""
        if m is not None and m.dtype == torch.uint8:
            mp, mpitch = _pitched(m, lead, (TW_CELLS,), torch.uint8)
            return mp, mpitch, flags | TW_F_MATRIX_CODE
        mp, mpitch = _pitched(m, lead, (TW_CELLS,), torch.float32)
        return mp, mpitch, flags

    @staticmethod
    def decode_matrix(codes):
        """uint8 code frames -> the float matrix of Env_transact.matrix_env (env_buffer.py:300-336)."""
        lut = torch.tensor(MATRIX_CODE_VALUES, dtype=torch.float32, device=codes.device)
        return lut[codes.long()]

    def _launch(self, fn, name, lead, T, actions, draws, out, flags):
        V = self.view_size
        op, opitch = _pitched(out.get("obs"), lead, (V, V, 3), torch.uint8)
        mp, mpitch, flags = self._matrix_arg(out.get("matrix"), lead, flags)
        for k, dt in (("pos", torch.float32), ("reward", torch.float32), ("terminated", torch.uint8),
                      ("truncated", torch.uint8)):
            t = out.get(k)
            assert t is None or (t.is_contiguous() and t.dtype == dt and tuple(t.shape[:len(lead)]) == lead), k
        args = [self._h] + ([T] if T is not None else []) + [
            _dense(actions, torch.int32), _ptr(draws), op, opitch, mp, mpitch, _ptr(out.get("pos")),
            _ptr(out.get("reward")), _ptr(out.get("terminated")), _ptr(out.get("truncated")), flags, self._stream()]
        _lib.check(fn(*args), name)
        return out

    def step(self, actions, out, draws=None, autoreset=False, policy_idx=False):
        flags = (TW_F_AUTORESET if autoreset else 0) | (TW_F_POLICY_IDX if policy_idx else 0)
        assert actions.shape == (self.num_envs,)
        if draws is not None:
            assert draws.shape == (self.num_envs, TW_DRAW_WORDS) and draws.dtype == torch.int32 and draws.is_contiguous()
        return self._launch(_lib.lib().tw_step, "tw_step", (self.num_envs,), None, actions, draws, out, flags)

    def rollout(self, T, out, actions=None, draws=None, autoreset=True, policy_idx=True):
        flags = (TW_F_AUTORESET if autoreset else 0) | (TW_F_POLICY_IDX if policy_idx else 0)
        if actions is not None:
            assert actions.shape == (T, self.num_envs)
        if draws is not None:
            assert draws.shape == (T, self.num_envs, TW_DRAW_WORDS) and draws.dtype == torch.int32 and draws.is_contiguous()
        return self._launch(_lib.lib().tw_rollout, "tw_rollout", (T, self.num_envs), T, actions, draws, out, flags)

    def fill_actions(self, T):
        a = torch.empty((T, self.num_envs), dtype=torch.int32, device=self.device)
        _lib.check(_lib.lib().tw_fill_actions(self._h, T, _ptr(a), self._stream()), "tw_fill_actions")
        return a

    def gen_obs(self, view_size=None):
        V = view_size or self.view_size
        obs = torch.empty((self.num_envs, V, V, 3), dtype=torch.uint8, device=self.device)
        _lib.check(_lib.lib().tw_gen_obs(self._h, V, _ptr(obs), 0, self._stream()), "tw_gen_obs")
        return obs

    def time_rollout(self, T, out, actions=None, autoreset=True, iters=10):
        """Mean kernel time (ms) of one tw_rollout launch, HIP events on the current stream."""
        ms = C.c_float()
        V = self.view_size
        lead = (T, self.num_envs)
        flags = (TW_F_AUTORESET if autoreset else 0) | TW_F_POLICY_IDX
        op, opitch = _pitched(out.get("obs"), lead, (V, V, 3), torch.uint8)
        mp, mpitch, flags = self._matrix_arg(out.get("matrix"), lead, flags)
        _lib.check(_lib.lib().tw_time_rollout(
            self._h, T, _dense(actions, torch.int32), op, opitch, mp, mpitch,
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
        keep = []

        def p(a, dt, shape):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dt)
            assert a.shape == shape
            keep.append(a)
            return a.ctypes.data_as(C.c_void_p)
        N = self.num_envs
        _lib.check(_lib.lib().tw_set_state_host(self._h, p(type_plane, np.uint8, (N, TW_CELLS)),
                                                p(colour_plane, np.uint8, (N, TW_CELLS)),
                                                p(records, np.int32, (N, TW_REC_WORDS))), "tw_set_state_host")

    @staticmethod
    def field(records, name, k=0):
        return records[..., FIELDS[name] + k]
