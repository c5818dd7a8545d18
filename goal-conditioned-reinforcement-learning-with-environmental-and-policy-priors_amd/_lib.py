"""ctypes binding of libtwoarmy_hip.so (C ABI declared in include/twoarmy.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load this module
raises, and every op built on it fails loudly.
"""
import ctypes as C
import os
import subprocess

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libtwoarmy_hip.so")
CSRC_DIR = os.path.join(_PKG_DIR, "csrc")

TW_CELLS = 289
TW_REC_WORDS = 48
TW_DRAW_WORDS = 8
TW_F_AUTORESET = 1
TW_F_POLICY_IDX = 2
TW_F_MATRIX_CODE = 4
TW_F_SLAB_HIPMALLOC = 8

# enum tw_field (include/twoarmy.h)
FIELDS = dict(AX=0, AY=1, DIR=2, STEP_COUNT=3, STEP_MOVE=4, PONE=5, PATROL=6, UP1=7, RIGHT2=8, UPD_LONG=9,
              UPD_HORIZ=10, RISK=11, FIRST_ROOM2=12, OBX=13, OBY=16, O1X=19, O1Y=22, O1_VALID=25, O2X=26,
              O2Y=30, O2_VALID=34, GOAL_X=35, GOAL_Y=36, T=37, ERROR=38, MAX_STEPS=39, EPISODES=40,
              LAST_REWARD=41, LAST_TERM=42, LAST_TRUNC=43, WALL_I1=44, WALL_I2=45)

ENV_ERRORS = {1: AttributeError, 2: AssertionError, 3: TypeError}


class TwoarmyLibraryError(RuntimeError):
    pass


def build(force=False):
    """Compile every HIP source for gfx950 into libtwoarmy_hip.so (in-tree)."""
    cmd = ["make", "-s", "-C", CSRC_DIR] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    return LIB_PATH


_vp = C.c_void_p


class TwOutputs(C.Structure):
    """struct tw_outputs (include/twoarmy.h)."""
    _fields_ = [("obs", _vp), ("matrix", _vp), ("pos", _vp), ("reward", _vp), ("terminated", _vp), ("truncated", _vp),
                ("obs_pitch", C.c_int), ("mat_pitch", C.c_int), ("T", C.c_int), ("n_envs", C.c_int), ("flags", C.c_int),
                ("device", C.c_int), ("backing", C.c_int), ("slab_bytes", C.c_uint64), ("slab", _vp)]


_SIGS = {
    "tw_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32]),
    "tw_destroy": (C.c_int, [_vp]),
    "tw_reset": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    # (handle, actions, draws, obs, obs_pitch, matrix, mat_pitch, pos, reward, term, trunc, flags, stream)
    "tw_step": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "tw_rollout": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "tw_set_envs_per_wave": (C.c_int, [_vp, C.c_int]),
    "tw_set_pipeline": (C.c_int, [_vp, C.c_int]),
    "tw_fallback_count": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "tw_fill_actions": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "tw_state_ptrs": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "tw_get_state_host": (C.c_int, [_vp, _vp, _vp, _vp]),
    "tw_set_state_host": (C.c_int, [_vp, _vp, _vp, _vp]),
    "tw_gen_obs": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp]),
    "tw_n_envs": (C.c_int, [_vp]),
    "tw_view_size": (C.c_int, [_vp]),
    "tw_last_hip_error": (C.c_int, []),
    "tw_version": (C.c_char_p, []),
    "tw_build_id": (C.c_char_p, []),
    "tw_last_error_message": (C.c_char_p, []),
    "tw_alloc_outputs": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(TwOutputs)]),
    "tw_free_outputs": (C.c_int, [C.POINTER(TwOutputs)]),
    "tw_abi_sizeof_outputs": (C.c_int, []),
    "tw_time_rollout": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp,
                                  C.POINTER(C.c_float)]),
}

_f, _i, _u64, _i64 = C.c_float, C.c_int, C.c_uint64, C.c_int64
_SIGS.update({
    # include/twoarmy_ppo.h
    "ppo_sample": (C.c_int, [_vp, _i, _i, _vp, _u64, _u64, _vp, _vp, _vp]),
    "ppo_sample_dev": (C.c_int, [_vp, _i, _i, _vp, _u64, _u64, _vp, _vp, _vp, _vp]),
    "ppo_gae": (C.c_int, [_vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ppo_adv_norm": (C.c_int, [_vp, _i64, _f, _vp, _vp]),
    "ppo_loss_fwd_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "ppo_loss_fwd_bwd_masked": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "ppo_gather_stack": (C.c_int, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "mg_gen_obs": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp]),
    "mg_step": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "ppo_her_relabel": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _u64, C.c_uint32, C.c_uint32, _i, _i, _i, _vp, _vp, _vp,
                                  _vp, _vp, _vp, _vp, _vp]),
    "ppo_her_relabel_window": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _u64, C.c_uint32, C.c_uint32, _i, _i, _i, _i, _vp,
                                         _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ppo_lstm_cell": (C.c_int, [_vp, _vp, C.c_longlong, _vp, _vp, _vp, _i, _i, _vp]),
    "ppo_decoder_frames": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp]),
    "ppo_gather_stack_u8": (C.c_int, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "ppo_age_scan": (C.c_int, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "ppo_bias_relu_nhwc": (C.c_int, [_vp, _vp, _i64, _i, _vp]),
    "ppo_conv1_up4_bias_relu": (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "ppo_conv1_up4_bias_relu_c": (C.c_int, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ppo_conv1_up4_bwd_groups": (C.c_int, [_i]),
    "ppo_conv1_up4_bwd": (C.c_int, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ppo_relu_bwd_bias_grad_nhwc_blocks": (C.c_int, [_i64, _i]),
    "ppo_relu_bwd_bias_grad_nhwc": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i, _vp]),
})

_lib = None


def exported_symbols():
    return sorted(_SIGS)


def lib():
    """Load the HIP library (never falls back to anything else)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TwoarmyLibraryError(
                "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64 (soname libamdhip64.so.7).  Import torch FIRST so the
        # dynamic loader resolves our DT_NEEDED libamdhip64.so.7 to that already-loaded runtime; loading
        # us first would pull /opt/rocm's copy and leave the process with two HIP runtimes (ours then
        # sees no devices: "invalid device ordinal").
        import torch  # noqa: F401
        try:
            h = C.CDLL(LIB_PATH)
        except OSError as ex:
            raise TwoarmyLibraryError("cannot load %s: %s" % (LIB_PATH, ex)) from ex
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(h, name)
            except AttributeError as ex:
                raise TwoarmyLibraryError("%s lacks symbol %s" % (LIB_PATH, name)) from ex
            fn.restype, fn.argtypes = res, args
        runtimes = set()
        try:
            with open("/proc/self/maps") as f:
                for line in f:
                    if "libamdhip64" in line:
                        runtimes.add(line.split()[-1])
        except OSError:
            pass
        if h.tw_abi_sizeof_outputs() != C.sizeof(TwOutputs):
            raise TwoarmyLibraryError("struct tw_outputs: the library has %d bytes, the ctypes mirror %d"
                                      % (h.tw_abi_sizeof_outputs(), C.sizeof(TwOutputs)))
        if len(runtimes) > 1:
            raise TwoarmyLibraryError("two HIP runtimes mapped (%s): import torch before loading %s"
                                      % (sorted(runtimes), LIB_PATH))
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        raise TwoarmyLibraryError("%s failed: rc=%d hipError=%d %s" % (
            what, rc, lib().tw_last_hip_error(), lib().tw_last_error_message().decode()))
