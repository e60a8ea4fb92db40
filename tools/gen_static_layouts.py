"""Developer tool: regenerates diffrl_amd/csrc/dsim_static_layouts.hpp -- the LDS layouts (offsets + sizes) of the shipped
models and of every user model under csrc/user_models/ as all-constexpr structs -- with the product's generator
(diffrl_amd/specialise.py, which calls the layout builder the library runs at dsim_model_create).  Run after a change of
dsim_layout.hpp.  Build-time constants only: no run-time code generation."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffrl_amd import specialise  # noqa: E402

if __name__ == "__main__":
    txt = specialise.render(specialise.shipped_templates() + specialise.user_templates())
    open(specialise.HEADER, "w").write(txt)
    print("wrote", specialise.HEADER, len(txt), "bytes")
