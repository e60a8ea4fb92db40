"""Compiles the original MJCF / URDF / SNU assets into builder snapshots (diffrl_amd/envs/assets/*.npz)
with diffrl_amd's own loaders, so that the environments can be constructed on machines that do not
have the asset files (e.g. the GPU box).  Run in the build container:  python tools/compile_assets.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffrl_amd import envs  # noqa: E402
from diffrl_amd.envs.dflex_env import ASSET_DIR, find_asset  # noqa: E402

if __name__ == "__main__":
    jobs = [("ant", envs.AntEnv, "ant.xml"), ("humanoid", envs.HumanoidEnv, "humanoid.xml"),
            ("cartpole", envs.CartPoleSwingUpEnv, "cartpole.urdf"), ("snu_humanoid", envs.SNUHumanoidEnv, "snu/human.xml"),
            ("hopper", envs.HopperEnv, "hopper.xml"), ("half_cheetah", envs.CheetahEnv, "half_cheetah.xml")]
    for name, cls, probe in jobs:
        assert find_asset(probe) is not None, "original asset %s not found (set DIFFRL_ASSETS)" % probe
        b = cls.make_builder()
        if isinstance(b, tuple):
            b = b[0]
        b.save(os.path.join(ASSET_DIR, name + ".npz"))
        print("compiled", name, "links:", len(b.joint_type), "shapes:", len(b.shape_body), "muscles:", len(b.muscle_start))
