"""The one wrapper of the reference's gym_minigrid/wrappers.py that lies on the path:
ViewSizeWrapper (wrappers.py:428-460), the only place a 7x7 egocentric view exists.  The reference re-slices the
grid with gen_obs_grid(agent_view_size) after every reset / step; here that is one tw_gen_obs launch at the requested
view size on the wrapped env's device state.  (The other wrappers -- bonuses, one-hot / RGB / flat / symbolic
observations -- are out of scope; a vector env takes its view size as a constructor argument instead, see
TwoarmyEngine / TwoarmyVecEnv.)"""
from .minigrid import _Space


class ViewSizeWrapper:
    def __init__(self, env, agent_view_size=7):
        assert agent_view_size % 2 == 1
        assert agent_view_size >= 3
        self.env = env
        self.unwrapped = getattr(env, "unwrapped", env)
        self.agent_view_size = agent_view_size
        self.observation_space = dict(getattr(env, "observation_space", {}))
        self.observation_space["image"] = _Space(shape=(agent_view_size, agent_view_size, 3))

    def __getattr__(self, name):                       # everything else is the wrapped env's (gym.Wrapper behaviour)
        return getattr(self.env, name)

    def observation(self, obs):
        grid, vis_mask = self.unwrapped.gen_obs_grid(self.agent_view_size)
        return {**obs, "image": grid.encode(vis_mask)}

    def reset(self, **kwargs):
        out = self.env.reset(**kwargs)
        if kwargs.get("return_info"):
            return self.observation(out[0]), out[1]
        return self.observation(out)

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        return self.observation(obs), reward, terminated, truncated, info
