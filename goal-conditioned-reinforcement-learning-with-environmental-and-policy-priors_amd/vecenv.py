"""gym.vector-style front end of the HIP engine: N Twoarmy envs stepped by one kernel launch.

  env = TwoarmyVecEnv("MiniGrid-twoarmy-17x17-v6", num_envs=4096)
  obs = env.reset()                                  # uint8 [N, V, V, 3] device tensor
  obs, reward, terminated, truncated, info = env.step(actions)      # actions: int tensor [N]

Semantics follow gym.vector with in-kernel auto-reset: for envs that finished, `obs` is the first
observation of the next episode and info["final_observation"] holds the terminal one (as a dense
tensor + info["_final_observation"] mask, no host sync).  `policy_actions=True` (default) takes the
policy's 5 indices (4 -> done) like Env_transact.env_action (reference soa/env_buffer.py:364-376).
Extra per-step tensors the reference computes in Python are fused into the same launch:
`env.state_matrix` [N,289] (matrix_env) and `env.agent_yx` [N,2] (data_env).
"""
import torch

from .engine import TwoarmyEngine

_IDS = {"MiniGrid-twoarmy-17x17-v4": 4, "MiniGrid-twoarmy-17x17-v6": 6, "v4": 4, "v6": 6, 4: 4, 6: 6}


class TwoarmyVecEnv:
    def __init__(self, env_id="MiniGrid-twoarmy-17x17-v6", num_envs=4096, agent_view_size=17, device=None,
                 seed=9981, env_id0=0, policy_actions=True, autoreset=True):
        self.variant = _IDS[env_id]
        self.num_envs = int(num_envs)
        self.view_size = agent_view_size
        self.policy_actions, self.autoreset = policy_actions, autoreset
        self.engine = TwoarmyEngine(self.variant, num_envs, agent_view_size, device=device, seed=seed, env_id0=env_id0)
        self.device = self.engine.device
        self._out = self.engine.alloc_outputs()
        self._init_obs = torch.empty((self.num_envs, agent_view_size, agent_view_size, 3), dtype=torch.uint8,
                                     device=self.device)
        self.engine.reset(obs=self._init_obs)                      # the reset observation is a constant of the task
        self.goal_yx = torch.tensor([2.0, 14.0], device=self.device).expand(self.num_envs, 2)
        self.single_observation_shape = (agent_view_size, agent_view_size, 3)
        self.single_action_n = 5 if policy_actions else 7

    def reset(self):
        self.engine.reset(obs=self._out["obs"])
        return self._out["obs"]

    def step(self, actions):
        a = actions.to(device=self.device, dtype=torch.int32).contiguous()
        o = self._out
        self.engine.step(a, o, autoreset=self.autoreset, policy_idx=self.policy_actions)
        done = (o["terminated"] | o["truncated"]).bool()
        info = {}
        obs = o["obs"]
        if self.autoreset:
            info["final_observation"] = obs
            info["_final_observation"] = done
            obs = torch.where(done.view(-1, 1, 1, 1), self._init_obs, obs)
        return obs, o["reward"], o["terminated"].bool(), o["truncated"].bool(), info

    @property
    def state_matrix(self):
        return self._out["matrix"]

    @property
    def agent_yx(self):
        return self._out["pos"]

    def close(self):
        self.engine.close()
