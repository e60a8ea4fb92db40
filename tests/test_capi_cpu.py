"""CPU-only checks of the C-ABI library: it loads and exports every symbol include/dsim.h declares.
(No compute calls: there is no GPU in the build container and no CPU fallback in the product.)"""
import ctypes
import os
import re

import pytest

import diffrl_amd.capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dsim.h")).read()
    return sorted(set(re.findall(r"\b(dsim_[a-z_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert sorted(capi.EXPORTS) == _declared()


def test_library_exports_all_symbols():
    if not os.path.exists(capi.LIB_PATH):
        pytest.skip("libdsim_hip.so not built yet (run __graft_entry__.build())")
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.dsim_version() >= 100


def test_modeldesc_matches_header_field_order():
    src = open(os.path.join(ROOT, "include", "dsim.h")).read()
    body = src[src.index("typedef struct dsim_model_desc {"):src.index("} dsim_model_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\b([a-zA-Z_0-9]+)(?:\[3\])?;", body)
    assert names == [f[0] for f in capi.ModelDesc._fields_]


def _struct_fields(name):
    src = open(os.path.join(ROOT, "include", "dsim.h")).read()
    body = src[src.index("typedef struct %s {" % name):src.index("} %s;" % name)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.split("{")[-1].strip()
        if not decl:
            continue
        for part in decl.split(","):
            out.append(re.sub(r"\[\d+\]", "", part.strip().split()[-1].lstrip("*")))
    return out


def test_envspec_and_episode_match_header_field_order():
    assert _struct_fields("dsim_env_spec") == [f[0] for f in capi.EnvSpec._fields_]
    assert _struct_fields("dsim_episode") == [f[0] for f in capi.Episode._fields_]
    # layout as the C compiler sees it
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(
            '#include <stdio.h>\n#include <stddef.h>\n#include "dsim.h"\nint main(void){printf("%zu %zu %zu %zu %zu\\n",'
            'sizeof(dsim_episode), offsetof(dsim_episode, noise_q), offsetof(dsim_episode, noise_angle),'
            'offsetof(dsim_episode, seed), sizeof(dsim_env_spec));return 0;}\n')
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        c = [int(x) for x in subprocess.check_output([os.path.join(d, "t")]).split()]
    E = capi.Episode
    assert c == [ctypes.sizeof(E), E.noise_q.offset, E.noise_angle.offset, E.seed.offset, ctypes.sizeof(capi.EnvSpec)]


def test_engine_refuses_cpu():
    from diffrl_amd.engine import Engine
    from oracle_lib import template_from_golden
    with pytest.raises(capi.DsimError):
        Engine(template_from_golden("cartpole"), "cpu")
