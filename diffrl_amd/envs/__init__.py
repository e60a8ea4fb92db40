from .ant import AntEnv  # noqa: F401
from .cartpole_swing_up import CartPoleSwingUpEnv  # noqa: F401
from .dflex_env import DFlexEnv  # noqa: F401
from .humanoid import HumanoidEnv  # noqa: F401
from .snu_humanoid import SNUHumanoidEnv  # noqa: F401
from .planar import CheetahEnv, HopperEnv  # noqa: F401
