/*
 * twoarmy.h -- C ABI of the MI355X-native MiniGrid-Twoarmy step/observation engine and the
 * PPO math kernels (libtwoarmy_hip.so, built from <package>/csrc/ by hipcc for gfx950).
 *
 * The reference (pure Python) has no FFI layer; its boundary for this path is the Python object
 * API of gym_minigrid.MiniGridEnv / Twoarmy_v{4,6} and soa.env_buffer / soa.agent.PPO.  Each
 * entry point below names the reference interface it replaces (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a maintainer adds on the
 * reference side.
 *
 * Conventions: plain pointers and sizes only; every data pointer is a DEVICE pointer owned by
 * the caller unless a function name ends in _host; `stream` is a hipStream_t passed as void*
 * (NULL = default stream); all launches are asynchronous on that stream; return value 0 = ok,
 * negative = error (TW_E_*); no exceptions cross the ABI; one handle per GPU; a handle is not
 * thread-safe.  Nullable output pointers skip that output.
 */
#ifndef TWOARMY_H
#define TWOARMY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TW_GRID 17            /* gym_minigrid/__init__.py:13,19  kwargs size=17 */
#define TW_CELLS 289
#define TW_REC_WORDS 48       /* int32 words per env in the scalar record (see enum tw_field) */
#define TW_DRAW_WORDS 8       /* uint32 draw words per env-step (see enum tw_slot) */

/* error returns */
#define TW_OK 0
#define TW_E_ARG (-1)         /* bad argument */
#define TW_E_HIP (-2)         /* HIP runtime error (tw_last_hip_error gives the hipError_t) */
#define TW_E_NOMEM (-3)

/* step flags */
#define TW_F_AUTORESET 1      /* reset a done env in place after its outputs are written
                                 (soa/train_ppo.py:104: every episode starts with reset()) */
#define TW_F_POLICY_IDX 2     /* actions are policy indices 0..4; 4 -> done(6)
                                 (soa/env_buffer.py:364-376 Env_transact.env_action) */
#define TW_F_MATRIX_CODE 4    /* reduced-precision frames (BASELINE config 5): `state_matrix` points to uint8 codes
                                 [N][mat_pitch BYTES] (native pitch 304) -- 0 free/goal (0.9), 1 wall (-0.9), 2 ball (-0.5),
                                 3 agent (0.3) -- instead of float[289]; exact (the matrix is a 4-entry LUT) and 4x smaller.
                                 ppo_gather_stack_u8 expands it back to fp32 policy inputs. */

#define TW_F_SLAB_HIPMALLOC 8 /* tw_alloc_outputs only: back the slab with plain hipMalloc instead of mapped 2 MiB chunks */

/* per-env scalar record, int32 words (AoS: one 192-byte record per env) */
enum tw_field {
    TW_AX = 0, TW_AY, TW_DIR,                 /* agent_pos, agent_dir (minigrid.py:927-928) */
    TW_STEP_COUNT, TW_STEP_MOVE,              /* minigrid.py:972, twoarmy_v6.py:15 */
    TW_PONE, TW_PATROL, TW_UP1, TW_RIGHT2, TW_UPD_LONG, TW_UPD_HORIZ, TW_RISK, TW_FIRST_ROOM2,
    TW_OBX = 13, TW_OBY = 16,                 /* obstacles[0..2].cur_pos */
    TW_O1X = 19, TW_O1Y = 22, TW_O1_VALID = 25,   /* obstacles1[0..2].cur_pos (None until spawn) */
    TW_O2X = 26, TW_O2Y = 30, TW_O2_VALID = 34,   /* obstacles2[0..3].cur_pos */
    TW_GOAL_X = 35, TW_GOAL_Y = 36,
    TW_T = 37,                                /* step() calls since creation: draw counter */
    TW_ERROR = 38,                            /* what the reference would have raised: enum tw_env_error */
    TW_MAX_STEPS = 39,
    TW_EPISODES = 40,                         /* finished episodes (statistics) */
    TW_LAST_REWARD = 41, TW_LAST_TERM = 42, TW_LAST_TRUNC = 43,
    TW_WALL_I1 = 44, TW_WALL_I2 = 45          /* y / x offset of the two dropped 2x2 wall blocks (valid while TW_PONE) */
};

enum tw_env_error { TW_ENV_OK = 0, TW_ENV_ATTRIBUTE = 1 /* env action 4/5: minigrid.py:1397 */,
                    TW_ENV_ASSERT = 2 /* Grid.get/set bounds: minigrid.py:599-607 */,
                    TW_ENV_TYPE = 3 /* cur_pos None: twoarmy_v4.py:122-124 */ };

/* draw slots: value = lo + word % n; word = Philox4x32-10(key=seed, ctr=(env_id, t, slot>>2, 'TWOA'))[slot&3] */
enum tw_slot { TW_S_GATE = 0 /* twoarmy_v4.py:117,149 range(10) */, TW_S_WALL1 = 1 /* :184 range(9,13) */,
               TW_S_WALL2 = 2 /* :190 range(6,10) */, TW_S_SPAWN = 3 /* :215 range(6,10) */,
               TW_S_COIN_A = 4 /* :303 range(2) */, TW_S_COIN_B = 5 /* :310 range(2) */,
               TW_S_ACTION = 8 /* step-only benchmark stream, % 5 */ };

typedef struct tw_engine tw_engine;

/* Replaces gym.make("MiniGrid-twoarmy-17x17-v{4,6}") + Twoarmy_v{4,6}.__init__ + the reset() it
 * ends with (soa/train_ppo.py:80-85, twoarmy_v6.py:10-37, minigrid.py:945) for n_envs
 * independent instances.  variant: 4 or 6.  view_size: odd, 3..17 (agent_view_size,
 * minigrid.py:874; wrappers.py:428-460 ViewSizeWrapper).  env_id0: global id of env 0 (sharding). */
int tw_create(tw_engine **out, int variant, int n_envs, int view_size, int device_id,
              uint64_t seed, uint32_t env_id0);
int tw_destroy(tw_engine *e);

/* MiniGridEnv.reset (minigrid.py:947-980): regenerate the grid, step_count = 0; Twoarmy flags are
 * NOT touched (twoarmy_v6.py has no reset override).  mask: uint8[n_envs] or NULL (= all).
 * obs (nullable): uint8[n_envs][V][V][3] first observation (only rows with mask set are written). */
int tw_reset(tw_engine *e, const uint8_t *mask, uint8_t *obs, int obs_pitch, void *stream);

/* One Twoarmy_v{4,6}.step for every env (twoarmy_v6.py:83-325 / twoarmy_v4.py:82-322, incl.
 * MiniGridEnv.step minigrid.py:1333-1441, gen_obs :1443-1496) fused with
 * Env_transact.matrix_env / data_env (soa/env_buffer.py:300-334).
 *   actions        int32[N]            env actions (or policy indices with TW_F_POLICY_IDX)
 *   draws          uint32[N][8]|NULL   explicit draw words; NULL -> Philox(seed, env_id, t, slot)
 *   obs            uint8[N][V][V][3]   obs["image"], indexed [x][y][c] (minigrid.py:757-770)
 *   state_matrix   float[N][289]       0.9 free/goal, -0.9 wall, -0.5 ball, 0.3 agent
 *   pos            float[N][2]         agent (y, x)
 *   reward         float[N]            {-0.01,-0.1,-0.9,0.2,0.9}
 *   terminated, truncated  uint8[N]
 * Pitches (MI355X layout): obs_pitch = bytes between consecutive envs' images (0 = dense V*V*3),
 * mat_pitch = floats between consecutive envs' matrices (0 = dense 289).  With a 16-byte aligned
 * base, obs_pitch % 16 == 0 and >= roundup(V*V*3, 16) (880 for V=17), mat_pitch % 4 == 0 and >= 292
 * the engine takes its register-packed dwordx4 path and also writes the pad bytes/floats (zeros);
 * any other layout (incl. dense) uses a slower, equally exact generic path.
 */
int tw_step(tw_engine *e, const int32_t *actions, const uint32_t *draws, uint8_t *obs, int obs_pitch,
            float *state_matrix, int mat_pitch, float *pos, float *reward, uint8_t *terminated,
            uint8_t *truncated, int flags, void *stream);

/* T consecutive steps in ONE launch (the rollout of soa/train_ppo.py:107-123 with a supplied
 * action stream).  All arrays are time-major [T][N][...].  actions NULL -> policy indices from the
 * Philox action slot (implies TW_F_POLICY_IDX).  Grid planes stay in LDS for the whole launch. */
int tw_rollout(tw_engine *e, int T, const int32_t *actions, const uint32_t *draws, uint8_t *obs,
               int obs_pitch, float *state_matrix, int mat_pitch, float *pos, float *reward,
               uint8_t *terminated, uint8_t *truncated, int flags, void *stream);

/* Envs per wavefront of the rollout kernel: 1, 2 or 4 (0 = auto from n_envs; also TW_ENVS_PER_WAVE). */
int tw_set_envs_per_wave(tw_engine *e, int envs_per_wave);

/* Rollouts of >= 8 steps with auto-reset, native layouts and Philox draws use the pipelined kernel
 * (one logic wave + 15 emission waves per 16 envs) with the sequential kernel as in-stream fallback;
 * enable = 0 forces the sequential kernel (also TW_PIPELINE=0). */
int tw_set_pipeline(tw_engine *e, int enable);

/* Statistic: how many of this engine's pipelined launches so far left normal play (illegal action, injected / drifted
 * state) and were re-run from their input state by the flag-gated sequential launch behind them.  0 in normal play.
 * Synchronises the device. */
int tw_fallback_count(tw_engine *e, int *count);

/* Fill int32[T][N] with the Philox action-slot policy indices the engine would use for its next
 * T steps (t counted from each env's current TW_T). */
int tw_fill_actions(tw_engine *e, int T, int32_t *actions, void *stream);

/* State injection / inspection (parity tests, checkpointing).  Device pointers to the engine's own
 * SoA planes uint8[N][289] (index y*17+x, values OBJECT_TO_IDX / COLOR_TO_IDX minigrid.py:40-67)
 * and the int32[N][TW_REC_WORDS] records. */
int tw_state_ptrs(tw_engine *e, uint8_t **type_plane, uint8_t **colour_plane, int32_t **records);
int tw_get_state_host(tw_engine *e, uint8_t *type_plane, uint8_t *colour_plane, int32_t *records);
int tw_set_state_host(tw_engine *e, const uint8_t *type_plane, const uint8_t *colour_plane,
                      const int32_t *records);

/* gen_obs_grid(view).encode() for the current state without stepping (minigrid.py:1443-1496). */
int tw_gen_obs(tw_engine *e, int view_size, uint8_t *obs, int obs_pitch, void *stream);

int tw_n_envs(const tw_engine *e);
int tw_view_size(const tw_engine *e);
int tw_last_hip_error(void);
const char *tw_last_error_message(void);
const char *tw_version(void);
/* First 16 hex digits of sha256(kernel sources + include/ headers) the library was compiled from (csrc/Makefile).
 * bench.py only reports a committed HBM-traffic counter profile whose recorded id equals this one. */
const char *tw_build_id(void);

/* Output buffers of one T-step rollout (or one step: T = 1) allocated by the engine in its native MI355X layout,
 * all carved out of ONE device slab.  This is the counterpart of the arrays the reference's rollout loop appends to
 * (soa/train_ppo.py:116-123 stack pushes, Buffer_gridworld.store env_buffer.py:68-77): every user of the library --
 * the Python trainer, the vector env, a C caller -- gets the same placement without probing candidates.
 * Frames and images use the RECORD layout: ONE stream of blocks, one per env-step,
 *   float frames   [T][N] x { float matrix[292] (289 + 3 zero pad) | uint8 image[R] }      R = roundup(V*V*3, 16)
 *                  matrix = slab, obs = slab + 1168, obs_pitch = 1168 + R bytes, mat_pitch = obs_pitch / 4 floats
 *                  (V = 17: R = 880, 2048-byte blocks, each written by two full-wave stores of eight whole 128-byte lines)
 *   code frames    [T][N] x { uint8 image[R] | uint8 codes[304] }  (TW_F_MATRIX_CODE)
 *                  obs = slab, matrix = slab + R, obs_pitch = mat_pitch = R + 304 bytes
 * tw_rollout recognises these pointers / pitches.  (Two separate streams, which the API still accepts, make the store
 * bandwidth depend on where the driver happens to place them: 0.178 ... 0.230 ms per 4096 x 128 launch; the record
 * stream is 0.19-0.20 ms on every allocation.)
 *   pos float[T][N][2], reward float[T][N], terminated / truncated uint8[T][N]   (dense)
 * `backing`: how the slab is backed -- 1 (default) = 2 MiB physical chunks (hipMemCreate) mapped into one virtual range,
 * 0 = hipMalloc (also the fallback when the runtime refuses the mapping calls).  Pass the struct's members to tw_step /
 * tw_rollout.  tw_free_outputs returns the slab's physical memory (the struct is zeroed) but keeps a mapped slab's VIRTUAL
 * address range reserved for the life of the process -- slab_bytes of address space per released slab (about 1 GiB at
 * 4096 envs x 128 steps), nothing else: on ROCm 7.2 a range that went through hipMemAddressFree and was handed out again
 * by hipMemAddressReserve is read wrongly by hipMemcpy device -> host (tools/vmm_reuse_repro.hip).  Allocate output slabs
 * once and reuse them; a process that allocates and frees slabs in a loop only spends address space. */
typedef struct tw_outputs {
    uint8_t *obs;
    void *matrix;
    float *pos;
    float *reward;
    uint8_t *terminated;
    uint8_t *truncated;
    int obs_pitch, mat_pitch;
    int T, n_envs, flags, device, backing;
    uint64_t slab_bytes;
    void *slab;
} tw_outputs;
int tw_alloc_outputs(tw_engine *e, int T, int flags, tw_outputs *out);
int tw_free_outputs(tw_outputs *out);
int tw_abi_sizeof_outputs(void);      /* sizeof(tw_outputs) of the built library: lets a binding check its struct mirror */

/* Time one launch of the engine kernel with hipEvents on `stream` (bench.py roofline leg):
 * runs tw_rollout `iters` times back-to-back and returns the mean kernel time in milliseconds. */
int tw_time_rollout(tw_engine *e, int T, const int32_t *actions, uint8_t *obs, int obs_pitch,
                    float *state_matrix, int mat_pitch, float *pos, float *reward, uint8_t *terminated,
                    uint8_t *truncated, int flags,
                    int iters, void *stream, float *ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* TWOARMY_H */
