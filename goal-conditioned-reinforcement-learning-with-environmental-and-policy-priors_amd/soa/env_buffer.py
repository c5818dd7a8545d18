"""Env adapter + replay buffer of the PPO path (reference soa/env_buffer.py).

`Env_transact` keeps the reference's method names and return shapes (matrix_env, data_env,
env_action, reset, step) on top of the N = 1 facade env.  `matrix_env` restates the reference's 289-cell Python loop
as one numpy expression over the facade's type plane and returns the reference's float64 values (valid in any state,
also right after `reset()`); the vector path (`TwoarmyVecEnv`, `VecPPOTrainer`) takes the same matrix as the HIP
kernel's fused fp32 output, checked equal in tests/test_stack_gpu.py.  `Buffer_gridworld` is the numpy ring buffer with the
reference's store() and hindsight relabelling her_func() (env_buffer.py:68-77, 101-143) -- host logic,
identical index arithmetic, checked against tests/golden/her.npz -- and the 9-frame window variants of the predictor /
self-orientation entry points: pre_store(), future_3_position_store(), pre_her_func(), pre_f_her_func()
(env_buffer.py:79-99, 145-280), checked against tests/golden/window_her.npz.  The vector trainers do not copy windows:
they keep one frame per step and relabel by index (ppo_her_relabel with `skip` = 4, see VecPPOTrainer.relabel); these
classes are the N = 1 API and the statement of what the index arithmetic has to equal.
"""
import numpy as np


class Buffer_gridworld:
    def __init__(self):
        self.name = None
        self.grid_size = None
        self.transition = None
        self.pre_transition = None
        self.buffer_capacity = 100000
        self.buffer_pre_capacity = 100000
        self.buffer = []
        self.pre_buffer = []
        self.fp_buffer = []
        self.counter = 0
        self.pre_counter = 0
        self.fp_counter = 0
        self.full = False
        self.pre_full = False
        self.fp_full = False
        self.epo_counter_start = 0
        self.epo_counter_end = 0

    @staticmethod
    def ppo_dtype(grid_size=17):
        """Record layout of soa/train_ppo.py:93-97."""
        return np.dtype([("s", np.float32, (5, grid_size ** 2)), ("a", np.int64, (1,)), ("p", np.float32, (5, 2)),
                         ("g", np.float32, (2,)), ("r", np.float32, (1,)), ("d", np.float32, (1,)),
                         ("a_logp", np.float32, (1,))])

    @staticmethod
    def window_dtype(grid_size=17, with_f=False):
        """9-frame window records of soa/train_ppo_predictor.py:105-108 (with_f: soa/train_SoA.py:113-116): frames and
        positions of steps i-3 .. i+5 of record i, and action / reward / done / log-prob (/ predicted displacement) of the
        five transitions i .. i+4; the update reads transition i (slot 0, frames 0..4)."""
        f = [("s", np.float64, (9, grid_size ** 2)), ("a", np.int64, (5, 1)), ("p", np.float64, (9, 2)),
             ("g", np.float64, (2,)), ("r", np.float64, (5, 1)), ("d", np.int64, (5, 1)), ("a_logp", np.float64, (5, 1))]
        return np.dtype(f + ([("f", np.float64, (5, 2))] if with_f else []))

    def store(self, transition):
        if self.counter >= self.buffer_capacity:
            self.counter, self.full = 0, True
        self.buffer[self.counter] = transition
        self.counter += 1
        if self.counter == self.buffer_capacity:
            self.counter, self.full = 0, True
        return self.full

    def pre_store(self, transition):
        """store() for the window records (env_buffer.py:90-99)."""
        if self.pre_counter >= self.buffer_pre_capacity:
            self.pre_counter, self.pre_full = 0, True
        self.pre_buffer[self.pre_counter] = transition
        self.pre_counter += 1
        if self.pre_counter == self.buffer_pre_capacity:
            self.pre_counter, self.pre_full = 0, True
        return self.pre_full

    def future_3_position_store(self, transition):
        """Ring of the windows the orientation head trains on (env_buffer.py:79-88; it shares the window capacity)."""
        if self.fp_counter >= self.buffer_pre_capacity:
            self.fp_counter, self.fp_full = 0, True
        self.fp_buffer[self.fp_counter] = transition
        self.fp_counter += 1
        if self.fp_counter == self.buffer_pre_capacity:
            self.fp_counter, self.fp_full = 0, True
        return self.fp_full

    def _window_her(self, newgoal_size_in, step_fields):
        """Hindsight relabelling of the window records of the episode [epo_counter_start, pre_counter).  Record i's
        newest frame (slot 8) is the state after step i + 4, so the candidates are the first visits among THOSE states;
        a pick `index` appends records 0..index with g := that state, reward 0.9 / done 1 on transition index + 4
        (slot 4 of record index), and four more windows that slide this transition down to slot 0 with the achieved
        state repeated behind it -- the same tail the entry points store at a real episode end."""
        cap = self.buffer_pre_capacity
        end = self.pre_counter - 1
        episode = self.pre_buffer[self.epo_counter_start:end + 1].copy()
        _, first_visit = np.unique(episode["p"][:, 8, 0:2], return_index=True, axis=0)
        k = min(newgoal_size_in, first_visit.size)
        if end - self.epo_counter_start + 1 > 0:
            for index in np.random.choice(first_visit, size=k, replace=False):
                if not (0 < index < cap):
                    continue
                seg = np.empty(index + 5, dtype=episode.dtype)
                seg[:index + 1] = episode[:index + 1]
                seg["g"][:index + 1] = episode["p"][index, 8, 0:2]
                seg["r"][index, 4] = 0.9
                seg["d"][index, 4] = 1
                last = seg[index]
                for j in range(index + 1, index + 5):
                    seg[j] = seg[j - 1]
                    for name in ("p", "s"):
                        seg[name][j] = np.concatenate([seg[name][j - 1][1:], last[name][8:9]])
                    for name in step_fields:
                        seg[name][j] = np.concatenate([seg[name][j - 1][1:], last[name][4:5]])
                n = index + 5
                if end + 1 + n <= cap:
                    self.pre_buffer[end + 1:end + 1 + n] = seg
                    end += n
                else:                                   # wrap around the ring
                    over = end + 1 + n - cap
                    self.pre_buffer[end + 1:cap] = seg[:n - over]
                    self.pre_buffer[:over] = seg[n - over:]
                    end = over - 1
                    self.pre_full = True
        self.epo_counter_end = end
        self.pre_counter = end + 1

    def pre_her_func(self, max_steps=50, newgoal_size_in=4):
        """env_buffer.py:145-209 (predictor entry point)."""
        self._window_her(newgoal_size_in, ("a", "r", "d", "a_logp"))

    def pre_f_her_func(self, max_steps=50, newgoal_size_in=4):
        """env_buffer.py:212-280 (self-orientation entry point: the windows also carry the `f` field)."""
        self._window_her(newgoal_size_in, ("a", "r", "d", "a_logp", "f"))

    def her_func(self, max_steps=50, newgoal_size_in=4):
        """Hindsight relabelling of the episode [epo_counter_start, counter): for up to 4 distinct
        first-visit positions (np.random.choice without replacement, global numpy RNG like the
        reference) append the prefix that reaches it with g := achieved (y,x), last r := 0.9, d := 1."""
        cap = self.buffer_capacity
        end = self.counter - 1
        episode = self.buffer[self.epo_counter_start:end + 1].copy()
        _, first_visit = np.unique(episode["p"][:, 4, 0:2], return_index=True, axis=0)
        k = min(newgoal_size_in, first_visit.size)
        if end - self.epo_counter_start + 1 > 0:
            for index in np.random.choice(first_visit, size=k, replace=False):
                if not (0 < index < cap):
                    continue
                prefix = episode[:index + 1].copy()
                prefix["g"][:] = prefix["p"][index, 4, 0:2]
                prefix["r"][index] = 0.9
                prefix["d"][index] = 1
                n = index + 1
                if end + 1 + n <= cap:
                    self.buffer[end + 1:end + 1 + n] = prefix
                    end += n
                else:                                   # wrap around the ring
                    over = end + 1 + n - cap
                    self.buffer[end + 1:cap] = prefix[:n - over]
                    self.buffer[:over] = prefix[n - over:]
                    end = over - 1
                    self.full = True
        self.epo_counter_end = end
        self.counter = end + 1


class Env_transact:
    def __init__(self):
        self.name = None
        self.grid = None
        self.size_agentob = 17 ** 2
        self.state_matrix = np.full((self.size_agentob,), 0.9)
        self.runstep = 0

    def matrix_env(self, env):
        """0.9 free / goal, -0.9 wall, -0.5 ball, 0.3 agent (env_buffer.py:300-318)."""
        self.grid = env.grid
        t = env.grid._t.reshape(-1)
        m = np.where(t == 2, -0.9, np.where(t == 6, -0.5, 0.9))
        i, j = env.agent_pos
        m[env.grid.height * j + i] = 0.3
        self.state_matrix = m
        return m

    def data_env(self, env):
        (i, j), (gi, gj) = env.agent_pos, env.goal_pos
        return np.array((j, i), dtype=float), np.array((gj, gi), dtype=float)

    def free_env(self, env):
        """(agent (y,x), passable span of the ball row, goal (y,x)) and its 10-deep stack (env_buffer.py:336-356; no caller
        in the reference's scripts -- kept for API completeness).  The span follows the first ball's column."""
        (i, j), (gi, gj) = env.agent_pos, env.goal_pos
        bx = env.obstacles[0].cur_pos[0]
        free = {6: (8, 9, 8, 10), 7: (8, 6, 8, 10)}.get(int(bx), (8, 6, 8, 7))
        state = np.concatenate((np.array((j, i), dtype=float), np.array(free), np.array((gj, gi), dtype=float)), axis=0)
        return state, np.tile(state, (10, 1))

    def pre_col(self, env):
        """Frame + its 8-deep stack (env_buffer.py:358-362; unused by the reference's scripts)."""
        m = self.matrix_env(env)
        return m, np.tile(m, (8, 1))

    def env_action(self, env, action_agent):
        a = env.actions
        return {0: a.left, 1: a.right, 2: a.up, 3: a.down, 4: a.done}.get(action_agent)

    def reset(self, env, window=None):
        env.reset()
        state_matrix = self.matrix_env(env)
        state, goal = self.data_env(env)
        return np.tile(state_matrix, (5, 1)), np.tile(state, (5, 1)), goal

    def predata_reset(self, env):
        """Initial 9-frame window: the reset state nine times (env_buffer.py:430-437)."""
        state, _ = self.data_env(env)
        return np.tile(self.matrix_env(env), (9, 1)), np.tile(state, (9, 1))

    def step(self, env, window, action, args=None):
        self.runstep += 1
        obs, reward, terminated, truncated, _ = env.step(action)
        done = 0
        if self.runstep > 49:
            truncated = True
        if terminated:
            done, reward = 1, 0.9
        return obs, reward, terminated, truncated, done
