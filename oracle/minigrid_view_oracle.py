"""CPU restatement of the reference's general MiniGridEnv observation (gym_minigrid/minigrid.py):
get_view_exts :1262-1293, Grid.slice :641-660, Grid.rotate_left :627-639, Grid.process_vis :795-832,
gen_obs_grid :1443-1478, Grid.encode :749-772 -- on encoded cells instead of WorldObj instances.

TEST INFRASTRUCTURE ONLY (checker for mg_gen_obs in <package>/csrc/minigrid_view.hip).
Pinned against tests/golden/occlusion.npz (images and visibility masks recorded from the reference's own
gen_obs / gen_obs_grid on random grids with every object class, all directions, with and without occlusion).

A cell is None (empty) or a (type, colour, state) triple, exactly what WorldObj.encode returns; the world grid
is given as the uint8 [W][H][3] array of Grid.encode() where (1, 0, 0) is an empty cell.
"""
import numpy as np

EMPTY, WALL, DOOR = 1, 2, 4
WALL_CELL = (2, 5, 0)                     # Wall().encode(): type wall, colour grey, state 0


def see_behind(cell):
    """WorldObj.see_behind :303-305 (True), Wall :422-423 (False), Door :439-440 (is_open <=> state 0)."""
    if cell is None:
        return True
    if cell[0] == WALL:
        return False
    if cell[0] == DOOR:
        return cell[2] == 0
    return True


class Grid:
    def __init__(self, width, height):
        self.width, self.height = width, height
        self.grid = [None] * (width * height)

    def get(self, i, j):
        assert 0 <= i < self.width and 0 <= j < self.height
        return self.grid[j * self.width + i]

    def set(self, i, j, v):
        assert 0 <= i < self.width and 0 <= j < self.height
        self.grid[j * self.width + i] = v

    @classmethod
    def from_encoded(cls, enc):
        W, H, _ = enc.shape
        g = cls(W, H)
        for i in range(W):
            for j in range(H):
                t, c, s = (int(v) for v in enc[i, j])
                g.set(i, j, None if t == EMPTY else (t, c, s))
        return g

    def rotate_left(self):
        grid = Grid(self.height, self.width)
        for i in range(self.width):
            for j in range(self.height):
                grid.set(j, grid.height - 1 - i, self.get(i, j))
        return grid

    def slice(self, topX, topY, width, height):
        grid = Grid(width, height)
        for j in range(height):
            for i in range(width):
                x, y = topX + i, topY + j
                if 0 <= x < self.width and 0 <= y < self.height:
                    v = self.get(x, y)
                else:
                    v = WALL_CELL
                grid.set(i, j, v)
        return grid

    def process_vis(self, agent_pos):
        mask = np.zeros((self.width, self.height), dtype=bool)
        mask[agent_pos[0], agent_pos[1]] = True
        for j in reversed(range(0, self.height)):
            for i in range(0, self.width - 1):
                if not mask[i, j]:
                    continue
                if not see_behind(self.get(i, j)):
                    continue
                mask[i + 1, j] = True
                if j > 0:
                    mask[i + 1, j - 1] = True
                    mask[i, j - 1] = True
            for i in reversed(range(1, self.width)):
                if not mask[i, j]:
                    continue
                if not see_behind(self.get(i, j)):
                    continue
                mask[i - 1, j] = True
                if j > 0:
                    mask[i - 1, j - 1] = True
                    mask[i, j - 1] = True
        for j in range(self.height):
            for i in range(self.width):
                if not mask[i, j]:
                    self.set(i, j, None)
        return mask

    def encode(self, vis_mask):
        array = np.zeros((self.width, self.height, 3), dtype=np.uint8)
        for i in range(self.width):
            for j in range(self.height):
                if vis_mask[i, j]:
                    v = self.get(i, j)
                    array[i, j] = (EMPTY, 0, 0) if v is None else v
        return array


def view_exts(ax, ay, agent_dir, V):
    if agent_dir == 0:
        return ax, ay - V // 2
    if agent_dir == 1:
        return ax - V // 2, ay
    if agent_dir == 2:
        return ax - V + 1, ay - V // 2
    if agent_dir == 3:
        return ax - V // 2, ay - V + 1
    raise AssertionError("invalid agent direction")


def gen_obs(enc, ax, ay, agent_dir, V, see_through_walls, carrying=None):
    """-> (image uint8[V][V][3], vis_mask bool[V][V]) of one env; `carrying` = None or a (type, colour, state)."""
    world = Grid.from_encoded(np.asarray(enc))
    topX, topY = view_exts(ax, ay, agent_dir, V)
    grid = world.slice(topX, topY, V, V)
    for _ in range(agent_dir + 1):
        grid = grid.rotate_left()
    if not see_through_walls:
        vis = grid.process_vis((V // 2, V - 1))
    else:
        vis = np.ones((V, V), dtype=bool)
    grid.set(V // 2, V - 1, tuple(carrying) if carrying is not None else None)
    return grid.encode(vis), vis


def gen_obs_batch(enc, ax, ay, agent_dir, V, see_through_walls, carrying=None):
    """Batched over N envs: enc [N][W][H][3]; carrying [N][3] with type 0 = nothing carried."""
    N = enc.shape[0]
    img = np.zeros((N, V, V, 3), np.uint8)
    vis = np.zeros((N, V, V), np.uint8)
    for n in range(N):
        c = None if carrying is None or int(carrying[n][0]) == 0 else tuple(int(v) for v in carrying[n])
        img[n], m = gen_obs(enc[n], int(ax[n]), int(ay[n]), int(agent_dir[n]), V, see_through_walls, c)
        vis[n] = m
    return img, vis


# ----------------------------------------------------------------------------- MiniGridEnv.step (base class)
DIR_TO_VEC = ((1, 0), (0, 1), (-1, 0), (0, -1))                 # minigrid.py:70-79
A_LEFT, A_RIGHT, A_UP, A_DOWN, A_DROP, A_TOGGLE, A_DONE = 0, 1, 2, 3, 4, 5, 6   # Actions, minigrid.py:849-864
E_ATTRIBUTE, E_ASSERTION = 1, 2


def can_overlap(cell):
    """WorldObj :291-293 (False); Goal :361, SubGoal :371, Floor :386, Lava :399 (True); Door :435-437 (is_open)."""
    if cell is None:
        return True
    t = cell[0]
    if t in (8, 11, 3, 9):
        return True
    if t == DOOR:
        return cell[2] == 0
    return False


def step(world, ax, ay, agent_dir, step_count, max_steps, action):
    """MiniGridEnv.step :1333-1441 without the final gen_obs.  `world` = Grid of encoded cells.
    -> (ax, ay, step_count, error, terminated, truncated, reward); on an exception the state is returned as far as
    the reference had mutated it (step_count already incremented) with error 1 = AttributeError (the action falls
    through to `self.actions.forward`, which the Actions enum lacks, :1397) or 2 = AssertionError (Grid.get out of
    range, :604-606)."""
    step_count += 1
    reward, terminated, truncated = 0.0, False, False

    def get(i, j):
        if not (0 <= i < world.width and 0 <= j < world.height):
            raise AssertionError
        return world.get(i, j)

    try:
        get(ax + DIR_TO_VEC[agent_dir][0], ay + DIR_TO_VEC[agent_dir][1])          # fwd_cell, :1341-1344
        moves = {A_LEFT: (-1, 0), A_RIGHT: (1, 0), A_UP: (0, -1), A_DOWN: (0, 1), A_DONE: (0, 0)}
        if action in moves:
            px, py = ax + moves[action][0], ay + moves[action][1]
            cell = get(px, py)
            if cell is None or can_overlap(cell):
                ax, ay = px, py
            if cell is not None and cell[0] == 8:
                terminated = True
                reward = 1 - 0.9 * (step_count / max_steps)                        # _reward(), :1061
        else:
            raise AttributeError                                                   # `self.actions.forward`, :1397
    except AttributeError:
        return ax, ay, step_count, E_ATTRIBUTE, False, False, 0.0
    except AssertionError:
        return ax, ay, step_count, E_ASSERTION, False, False, 0.0
    if step_count >= max_steps:
        truncated = True
    return ax, ay, step_count, 0, terminated, truncated, reward
