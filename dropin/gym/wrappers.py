"""gym.wrappers names that rl_games imports at module level for environments that are not DFlexEnv (never instantiated here)."""
from . import ObservationWrapper


class FlattenObservation(ObservationWrapper):
    pass


class FilterObservation(ObservationWrapper):
    pass


class TimeLimit(ObservationWrapper):
    pass
