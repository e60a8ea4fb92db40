"""The stand-in itself (see __init__.py for when it is used)."""
from . import spaces  # noqa: F401


class Env:
    metadata, reward_range, action_space, observation_space = {}, (-float("inf"), float("inf")), None, None

    def step(self, action):
        raise NotImplementedError

    def reset(self, **kw):
        raise NotImplementedError

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space, self.observation_space = getattr(env, "action_space", None), getattr(env, "observation_space", None)

    def __getattr__(self, name):
        return getattr(self.env, name)

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kw):
        return self.env.reset(**kw)

    @property
    def unwrapped(self):
        return self.env.unwrapped


class ObservationWrapper(Wrapper):
    pass


class RewardWrapper(Wrapper):
    pass


class ActionWrapper(Wrapper):
    pass


def make(*a, **kw):
    raise RuntimeError("gym stand-in (dropin/gym): gym.make is not available; only DFlexEnv environments run here")


class _Registry:
    """gym.envs.register(...) calls of rl_games' bundled test environments: recorded, never instantiated"""
    registered = {}

    def register(self, id=None, **kw):
        self.registered[id] = kw


envs = _Registry()
register = envs.register
