"""diffrl_amd -- MI355X-native differentiable articulated rigid-body simulation (DFlexEnv drop-in)."""
__version__ = "0.1.0"
