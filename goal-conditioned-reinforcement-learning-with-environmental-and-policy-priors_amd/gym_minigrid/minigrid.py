"""MiniGridEnv facade (N = 1) over the HIP engine -- API-compatible with what the reference's callers
touch (gym_minigrid/minigrid.py:849-980, 1333-1496; soa/env_buffer.py:300-334, 364-376, 413-461).

One `step()` = one tw_step launch on the GPU followed by a host read-back of the 1-env state, so this
class is plumbing (BASELINE config 0), not the fast path; thousands of envs go through
`twoarmy_amd.vecenv.TwoarmyVecEnv`.  There is no CPU implementation behind it.
"""
from enum import IntEnum

import numpy as np
import torch

from ..engine import TwoarmyEngine
from .._lib import ENV_ERRORS, FIELDS

OBJECT_TO_IDX = {"unseen": 0, "empty": 1, "wall": 2, "floor": 3, "door": 4, "key": 5, "ball": 6, "box": 7,
                 "goal": 8, "lava": 9, "agent": 10, "subgoal": 11}
IDX_TO_OBJECT = {v: k for k, v in OBJECT_TO_IDX.items()}
COLOR_TO_IDX = {"red": 0, "green": 1, "blue": 2, "purple": 3, "yellow": 4, "grey": 5}
IDX_TO_COLOR = {v: k for k, v in COLOR_TO_IDX.items()}


class WorldObj:
    """Read-only view of one grid cell / obstacle (type, color, cur_pos)."""

    def __init__(self, type_idx, color_idx, pos=None):
        self.type = IDX_TO_OBJECT[int(type_idx)]
        self.color = IDX_TO_COLOR[int(color_idx)]
        self.cur_pos = pos
        self.init_pos = pos

    def encode(self):
        return (OBJECT_TO_IDX[self.type], COLOR_TO_IDX[self.color], 0)

    def can_overlap(self):
        return self.type in ("goal", "subgoal", "floor", "lava")


class Grid:
    """Snapshot of the engine's SoA planes with the reference Grid's read API (minigrid.py:555-772)."""

    def __init__(self, type_plane, colour_plane, width=17, height=17):
        self.width, self.height = width, height
        self._t = np.asarray(type_plane, np.uint8).reshape(height, width)
        self._c = np.asarray(colour_plane, np.uint8).reshape(height, width)
        self.grid = [None if t == 1 else WorldObj(t, c, (k % width, k // width))
                     for k, (t, c) in enumerate(zip(self._t.reshape(-1), self._c.reshape(-1)))]

    def get(self, i, j):
        assert 0 <= i < self.width and 0 <= j < self.height
        return self.grid[j * self.width + i]

    def encode(self, vis_mask=None):
        out = np.zeros((self.width, self.height, 3), np.uint8)
        out[:, :, 0] = self._t.T
        out[:, :, 1] = self._c.T
        if vis_mask is not None:
            out[~np.asarray(vis_mask, bool)] = 0
        return out


class _Space:
    def __init__(self, n=None, shape=None):
        self.n, self.shape = n, shape


class MiniGridEnv:
    class Actions(IntEnum):          # minigrid.py:849-864 (absolute moves; 4/5 are not usable)
        left = 0
        right = 1
        up = 2
        down = 3
        drop = 4
        toggle = 5
        done = 6

    variant = 6

    def __init__(self, size=17, agent_pos=(3, 15), goal_pos=(14, 2), agent_view_size=17, max_steps=50, tile_size=32,
                 device=None, seed=9981, env_id=0, **kwargs):
        if size != 17 or tuple(agent_pos) != (3, 15) or tuple(goal_pos) != (14, 2) or max_steps != 50:
            raise NotImplementedError("the HIP engine implements the registered configuration: size=17, "
                                      "agent_pos=(3,15), goal_pos=(14,2), max_steps=50")
        self.width = self.height = size
        self.max_steps = max_steps
        self.agent_view_size = agent_view_size
        self.see_through_walls = True
        self.tile_size = tile_size
        self.actions = MiniGridEnv.Actions
        self.action_space = _Space(n=len(self.actions))
        self.observation_space = {"image": _Space(shape=(agent_view_size, agent_view_size, 3))}
        self.mission = "get to the green goal square"
        self.carrying = None
        self._eng = TwoarmyEngine(self.variant, 1, agent_view_size, device=device, seed=seed, env_id0=env_id)
        self._out = self._eng.alloc_outputs()
        self._act = torch.zeros(1, dtype=torch.int32, device=self._eng.device)
        self._refresh()

    # ---------------------------------------------------------------- state mirror
    def _refresh(self):
        ty, co, rec = self._eng.get_state()
        self._rec = rec[0]
        self.grid = Grid(ty[0], co[0])
        r = self._rec
        self.agent_pos = (int(r[FIELDS["AX"]]), int(r[FIELDS["AY"]]))
        self.agent_dir = int(r[FIELDS["DIR"]])
        self.goal_pos = (int(r[FIELDS["GOAL_X"]]), int(r[FIELDS["GOAL_Y"]]))
        self.step_count = int(r[FIELDS["STEP_COUNT"]])
        for name, key in (("step_move", "STEP_MOVE"), ("pone", "PONE"), ("patrol", "PATROL"), ("up1", "UP1"),
                          ("right2", "RIGHT2"), ("Update_longitudinal", "UPD_LONG"),
                          ("Update_horizontal", "UPD_HORIZ"), ("risk_count", "RISK"),
                          ("first_to_room2", "FIRST_ROOM2")):
            v = int(r[FIELDS[key]])
            setattr(self, name, v if name in ("step_move", "risk_count") else bool(v))

        def objs(xk, yk, n, valid=True):
            return [WorldObj(6, 4, (int(r[FIELDS[xk] + i]), int(r[FIELDS[yk] + i])) if valid else None)
                    for i in range(n)]
        self.obstacles = objs("OBX", "OBY", 3)
        self.obstacles1 = objs("O1X", "O1Y", 3, bool(r[FIELDS["O1_VALID"]]))
        self.obstacles2 = objs("O2X", "O2Y", 4, bool(r[FIELDS["O2_VALID"]]))

    # ---------------------------------------------------------------- gym API
    def reset(self, *, seed=None, return_info=False, options=None):
        obs = torch.empty((1, self.agent_view_size, self.agent_view_size, 3), dtype=torch.uint8,
                          device=self._eng.device)
        self._eng.reset(obs=obs)
        self._refresh()
        o = {"image": obs[0].cpu().numpy(), "direction": self.agent_dir, "mission": self.mission}
        return (o, {}) if return_info else o

    def step(self, action, draws=None):
        self._act[0] = int(action)
        d = None
        if draws is not None:
            d = torch.from_numpy(np.ascontiguousarray(draws, np.uint32).view(np.int32)).to(self._eng.device).view(1, 8)
        self._eng.step(self._act, self._out, draws=d)
        self._refresh()
        err = int(self._rec[FIELDS["ERROR"]])
        if err:
            raise ENV_ERRORS[err]("reference-equivalent error for env action %r" % (action,))
        self.state_matrix = self._out["matrix"][0].cpu().numpy()
        obs = {"image": self._out["obs"][0].cpu().numpy(), "direction": self.agent_dir, "mission": self.mission}
        return (obs, float(np.float64(round(float(self._out["reward"][0]), 2))), bool(self._out["terminated"][0]),
                bool(self._out["truncated"][0]), {})

    def gen_obs_grid(self, agent_view_size=None):
        V = agent_view_size or self.agent_view_size
        img = self._eng.gen_obs(V)[0].cpu().numpy()
        g = Grid(img[:, :, 0].T.reshape(-1), img[:, :, 1].T.reshape(-1), V, V)
        return g, np.ones((V, V), bool)

    def gen_obs(self):
        g, _ = self.gen_obs_grid()
        return {"image": g.encode(), "direction": self.agent_dir, "mission": self.mission}

    def get_full_render(self, *a, **k):
        """The reference renders 17x17 tiles per step and discards the image when server=True
        (env_buffer.py:456-459); rendering is out of scope."""
        return None

    def close(self):
        self._eng.close()
