/*
 * minigrid_view.h -- C ABI of the general MiniGrid observation kernel (libtwoarmy_hip.so,
 * <package>/csrc/minigrid_view.hip).  SURVEY.md section 8 row f2: the agent view of ANY MiniGridEnv subclass,
 * not only Twoarmy (whose fused step+view path is twoarmy.h).
 *
 * Replaces, for N envs resident in HBM, the reference's (paths relative to the reference root)
 *   MiniGridEnv.gen_obs / gen_obs_grid   gym_minigrid/minigrid.py:1443-1496
 *   get_view_exts :1262-1293, Grid.slice :641-660 (out of bounds -> Wall), Grid.rotate_left :627-639
 *   (applied agent_dir + 1 times), Grid.process_vis :795-832 (occlusion by walls and closed / locked doors,
 *   when see_through_walls is False), the carried object placed on the agent's cell :1469-1476,
 *   Grid.encode(vis_mask) :749-772 (invisible cells -> (0,0,0), empty cells -> (1,0,0)).
 *
 * World state = structure-of-arrays planes, one byte per cell, cell (x, y) at index y*W + x exactly like
 * Grid.grid (minigrid.py:562-569): `type` = OBJECT_TO_IDX (1 empty ... 11 subgoal; 0 is treated as empty like
 * WorldObj.decode does), `colour` = COLOR_TO_IDX, `state` = door state (0 open, 1 closed, 2 locked; NULL = all 0).
 *
 * Conventions as in twoarmy.h: device pointers, caller-owned, `stream` = hipStream_t as void*, asynchronous,
 * 0 = ok / negative = TW_E_*.
 */
#ifndef MINIGRID_VIEW_H
#define MINIGRID_VIEW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_MAX_VIEW 31          /* view sizes 1..31 (the reference's ViewSizeWrapper asks for odd sizes >= 3) */

/* type, colour, state: uint8[N][H*W] (state nullable); agent_x, agent_y, agent_dir: int32[N] (dir 0 right, 1 down,
 * 2 left, 3 up); carrying: uint8[N][3] = WorldObj.encode() of the carried object, type 0 = nothing (nullable);
 * image: uint8[N][image_pitch] with the [V][V][3] observation (indexed [i][j][channel] like obs["image"]) in the
 * first V*V*3 bytes of each row (image_pitch 0 = dense); vis_mask: uint8[N][V*V] indexed [i][j] (nullable).
 * The planes are read as 4-byte-aligned words: the words that hold the first and the last byte of a plane array are
 * read whole, i.e. up to 3 bytes before its start / after its end inside the same aligned word (always inside the
 * caller's allocation when that starts and ends on 4-byte boundaries, as hipMalloc / torch allocations do). */
int mg_gen_obs(const uint8_t *type, const uint8_t *colour, const uint8_t *state, int n_envs, int width, int height,
               const int32_t *agent_x, const int32_t *agent_y, const int32_t *agent_dir, const uint8_t *carrying,
               int view_size, int see_through_walls, uint8_t *image, int image_pitch, uint8_t *vis_mask, void *stream);

/* MiniGridEnv.step of the base class (minigrid.py:1333-1441) without the observation (call mg_gen_obs next):
 * step_count += 1; the cell in front of the agent is fetched first (an out-of-range front cell is the reference's
 * AssertionError even for sideways moves); actions 0 left, 1 right, 2 up, 3 down, 6 done(stay) are absolute moves
 * onto empty cells or objects with can_overlap() (goal, subgoal, floor, lava, open door); reaching a goal ->
 * terminated and reward = 1 - 0.9 * step_count / max_steps (_reward, :1061, in double like the Python float);
 * truncated = step_count >= max_steps.  Any other action value reaches `self.actions.forward`, which the
 * reference's Actions enum does not define (:1397).  The world planes are never modified (drop / toggle are
 * unreachable in the reference).
 *   error int32[N] (nullable): 0 ok, 1 AttributeError (action outside {0,1,2,3,6}), 2 AssertionError (Grid.get out
 *   of range); on an error the env keeps the mutation the reference had made (step_count) and reports reward 0,
 *   terminated = truncated = 0.  agent_x / agent_y / step_count are updated in place. */
int mg_step(const uint8_t *type, const uint8_t *state, int n_envs, int width, int height, const int32_t *action,
            int32_t *agent_x, int32_t *agent_y, const int32_t *agent_dir, int32_t *step_count, int max_steps,
            double *reward, uint8_t *terminated, uint8_t *truncated, int32_t *error, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MINIGRID_VIEW_H */
