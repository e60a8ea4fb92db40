"""One command from a user model to its own specialised kernel set:

    python -m diffrl_amd.specialise my_robot.npz --name MyRobot          # template saved with ArticulationTemplate.save()

A model whose LDS layout matches none of the compiled tables runs the GENERIC kernels (run-time offsets and sizes in SGPRs:
same results, about half the speed).  The specialised kernels are the same phase code instantiated with the model's offsets
and sizes as compile-time constants.  This module
  1. computes the model's layout with the builder the library itself runs at dsim_model_create (csrc/dsim_layout.hpp, here as
     a host-only shared object: no GPU needed),
  2. keeps the template under csrc/user_models/<Name>.npz (so the generated header is reproducible from inputs),
  3. regenerates csrc/dsim_static_layouts.hpp: the six shipped models + every user model, and
  4. rebuilds csrc/libdsim_hip.so with hipcc (the flags of __graft_entry__.build(); ~40 s per model).
At run time dsim_model_create compares the layout it builds with the tables and uses a specialised set only on an exact match;
`dsim_model_variant(m) > 0` (Engine.variant) confirms it.  Build-time constants only: nothing is generated or compiled at run
time -- unless asked for: Engine(..., specialise=True) / DSIM_AUTO_SPECIALISE=1 runs steps 1, 3 and 4 for the one model into
a library of its own on first use (ensure_library below; csrc/user_libs/, keyed by layout and source hash).  (tools/gen_static_layouts.py, the developer tool that regenerates the shipped tables, calls render() below.)
"""
import argparse
import shutil
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np

from . import capi
from .template import ArticulationTemplate

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
USER_DIR = os.path.join(CSRC, "user_models")
HEADER = os.path.join(CSRC, "dsim_static_layouts.hpp")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-fno-slp-vectorize",
               "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
DIMS = "L nq nd C M W NS D flags tmask".split()
PMASK_N = 10
TRUNK_MAX, TRUNK_CH = 6, 4
# DsimDims behind pmask, struct order: (name, length); length 0 = scalar (csrc/dsim_layout.hpp)
EXTRA = [("NT", 0), ("NLT", 0), ("LCAP", 0), ("CCAP", 0), ("trunk", TRUNK_MAX), ("tr_par", TRUNK_MAX), ("tr_nch", TRUNK_MAX),
         ("tr_ch", TRUNK_MAX * TRUNK_CH), ("tr_cb0", TRUNK_MAX), ("tr_ncb", TRUNK_MAX), ("tr_d0", TRUNK_MAX), ("tr_nd", TRUNK_MAX),
         ("MK", 0), ("pident", 0), ("RT_N", 0), ("CBMAX", 0), ("rt_lvl", 16), ("rt_d", 16), ("rt_kind", 16), ("DSH_OK", 0), ("DSH", 0), ("ND_ROOT", 0), ("SDMAX", 0), ("ADMAX", 0), ("JW_OK", 0), ("JW_FREE_ROOT", 0)]
_host = None


def _host_lib():
    """csrc/libdsim_layout_host.so, built on first use (g++, about a second)"""
    global _host
    if _host is None:
        so, src = os.path.join(CSRC, "libdsim_layout_host.so"), os.path.join(CSRC, "dsim_layout_host.cpp")
        deps = [src, os.path.join(CSRC, "dsim_layout.hpp"), os.path.join(os.path.dirname(HERE), "include", "dsim.h")]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps if os.path.exists(d)):
            # (two processes may get here together -- build() runs two specialise jobs side by side: each compiles to a name of its
            # own and renames it into place, so that nobody ever loads a half-written file)
            tmp = so + ".tmp%d" % os.getpid()
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", src, "-o", tmp])
            os.replace(tmp, so)
        _host = C.CDLL(so)
    return _host


def off_names():
    """field names of struct DsimOff, in order (parsed from the header the kernels are compiled with)"""
    src = open(os.path.join(CSRC, "dsim_layout.hpp")).read()
    body = src[src.index("struct DsimOff {"):]
    body = re.sub(r"//.*", "", body[:body.index("};")])
    names = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if stmt.startswith("struct"):
            stmt = stmt[stmt.index("{") + 1:].strip()
        if stmt.startswith("int "):
            names += [n.strip() for n in stmt[4:].split(",")]
    return names


def layout(t):
    """(offsets: dict name -> words, dims: dict) of template t, from the library's own layout builder"""
    desc, keep = capi.make_desc(t)
    n_off, err = C.c_int(0), C.c_char_p()
    buf = np.zeros(1024, np.int32)
    n = _host_lib().dsim_layout_dump(C.byref(desc), buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size), C.byref(n_off), C.byref(err))
    if n < 0:
        raise capi.DsimError("the layout builder refuses this model: %s" % (err.value or b"?").decode())
    names = off_names()
    assert n_off.value == len(names) and n <= buf.size, (n_off.value, len(names), n)
    dims = buf[n_off.value:n]
    d = dict(zip(DIMS, dims.tolist()))
    d["pmask"] = dims[len(DIMS):len(DIMS) + PMASK_N].tolist()
    pos = len(DIMS) + PMASK_N
    for name, cnt in EXTRA:
        if cnt == 0:
            d[name] = int(dims[pos]); pos += 1
        else:
            d[name] = dims[pos:pos + cnt].tolist(); pos += cnt
    assert pos == len(dims), "DsimDims has fields this module does not know: update EXTRA (%d of %d ints)" % (pos, len(dims))
    return dict(zip(names, buf[:n_off.value].tolist())), d


def shipped_templates():
    """(tag, template) of the six environments of the package, in the order of the shipped header"""
    from . import envs
    out = []
    for tag, cls in [("Cartpole", "CartPoleSwingUpEnv"), ("Ant", "AntEnv"), ("Humanoid", "HumanoidEnv"), ("Snu", "SNUHumanoidEnv"),
                     ("Hopper", "HopperEnv"), ("Cheetah", "CheetahEnv")]:
        out.append((tag, getattr(envs, cls)(num_envs=1, device="cpu", no_grad=True).model.template()))
    return out


def user_templates(user_dir=USER_DIR):
    if not os.path.isdir(user_dir):
        return []
    return [(f[:-4], ArticulationTemplate.load(os.path.join(user_dir, f))) for f in sorted(os.listdir(user_dir)) if f.endswith(".npz")]


def flat_table(off, dims):
    """the ints dsim_model_create compares (match_variant): DsimOff then DsimDims, struct order"""
    names = off_names()
    flat = [off[n] for n in names] + [dims[n] for n in DIMS] + list(dims["pmask"])
    for name, n in EXTRA:
        flat += [dims[name]] if n == 0 else list(dims[name])
    return flat


def render(models):
    """text of dsim_static_layouts.hpp for [(tag, template)]"""
    names = off_names()
    lines = ["// GENERATED by diffrl_amd/specialise.py (tools/gen_static_layouts.py) from dsim_layout.hpp -- do not edit by hand.",
             "// Per-model compile-time LDS layouts; see the generator's docstring.", "#pragma once", ""]
    tags = []
    for tag, t in models:
        if not re.fullmatch(r"[A-Za-z][A-Za-z0-9]*", tag):
            raise ValueError("model name %r: letters and digits only (it becomes part of C++ identifiers)" % tag)
        if tag in tags:
            raise ValueError("two models named %s" % tag)
        off, dims = layout(t)
        tags.append(tag)
        lines.append("struct DsimOff%s {" % tag)
        lines.append("    static constexpr int " + ", ".join("%s = %d" % (n, off[n]) for n in names) + ";")
        lines.append("};")
        lines.append("struct DsimDims%s {" % tag)
        lines.append("    static constexpr int " + ", ".join("%s = %d" % (n, dims[n]) for n in DIMS) + ";")
        lines.append("    static constexpr int pmask[%d] = {%s};" % (PMASK_N, ", ".join(str(v) for v in dims["pmask"])))
        for name, n in EXTRA:
            v = dims[name]
            if n == 0:
                lines.append("    static constexpr int %s = %d;" % (name, v))
            else:
                lines.append("    static constexpr int %s[%d] = {%s};" % (name, n, ", ".join(str(x) for x in v)))
        lines.append("};")
        lines.append("static const int kDsimStatic%s[] = {%s};" % (tag, ", ".join(str(v) for v in flat_table(off, dims))))
        lines.append("")
    lines.append("#ifndef DSIM_STATIC_VARIANTS  // (developer builds compile a subset: -D'DSIM_STATIC_VARIANTS(X)=X(Ant)')")
    lines.append("#define DSIM_STATIC_VARIANTS(X) " + " ".join("X(%s)" % t for t in tags))
    lines.append("#endif")
    return "\n".join(lines) + "\n"


def matches(t, header_text):
    """name of the table in `header_text` that template t's layout equals exactly (what dsim_model_create will pick), or None"""
    flat = flat_table(*layout(t))
    for m in re.finditer(r"static const int kDsimStatic(\w+)\[\] = \{([^}]*)\};", header_text):
        if [int(x) for x in m.group(2).split(",")] == flat:
            return m.group(1)
    return None


def build_library(out, header=None, only=None, quiet=False):
    """hipcc: csrc/dsim_hip.hip -> out.  header: another generated layouts header than csrc/dsim_static_layouts.hpp;
    only: list of model names to compile specialised sets for (the generic kernels are always there);
    quiet: capture the compiler's output (it is in the CalledProcessError if the build fails)"""
    cmd = ["hipcc"] + HIPCC_FLAGS
    if header:
        cmd.append('-DDSIM_STATIC_LAYOUTS_FILE="%s"' % os.path.abspath(header))
    if only:
        cmd.append("-DDSIM_STATIC_VARIANTS(X)=" + " ".join("X(%s)" % m for m in only))
    cmd += [os.path.join(CSRC, "dsim_hip.hip"), "-o", out]
    if quiet:
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    else:
        subprocess.check_call(cmd)
    return out


USER_LIBS = os.path.join(CSRC, "user_libs")


def source_hash():
    """hash of the kernel sources a library is built from (the layout tables excepted: a user library brings its own)"""
    import hashlib
    h = hashlib.sha1()
    for f in ("dsim_core.hpp", "dsim_hip.hip", "dsim_math.hpp", "dsim_layout.hpp", "dsim_literal.hpp", os.path.join("..", "..", "include", "dsim.h")):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:12]


STALE_DAYS = 30
_background = {}   # layout name -> threading.Thread of a compile in flight (ensure_library_background)


def cached_library(t, cache_dir=None):
    """path of the library ensure_library would return WITHOUT compiling anything, or None; never raises"""
    import hashlib
    try:
        if os.environ.get("DSIM_LIB") is None and matches(t, open(HEADER).read()):
            return capi.LIB_PATH
        flat = flat_table(*layout(t))
        name = "U" + hashlib.sha1(",".join(str(v) for v in flat).encode()).hexdigest()[:12]
        src = hashlib.sha1((source_hash() + "|" + " ".join(HIPCC_FLAGS)).encode()).hexdigest()[:10]
        out = os.path.join(cache_dir or os.environ.get("DSIM_USER_LIBS") or USER_LIBS, "libdsim_%s_%s.so" % (name, src))
        return out if os.path.exists(out) else None
    except Exception:
        return None


class _Job:
    """a background compile in flight: a thread of this process, or a DETACHED child process (join / is_alive of either)"""
    def __init__(self, thread=None, proc=None):
        self.thread, self.proc = thread, proc
        self.daemon = True

    def is_alive(self):
        return self.thread.is_alive() if self.thread is not None else self.proc.poll() is None

    def join(self, timeout=None):
        if self.thread is not None:
            self.thread.join(timeout)
        else:
            try:
                self.proc.wait(timeout)
            except subprocess.TimeoutExpired:
                pass


def _spawn_thread(t, key, cache_dir, log):
    """the compile in a daemon thread of this process (tests; dies with the process)"""
    import threading

    def work():
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("always")
            try:
                ensure_library(t, cache_dir, log)
            except Exception as ex:   # (a background convenience must never take the process down)
                print("[diffrl_amd.specialise] background compile failed: %s" % ex, file=sys.stderr, flush=True)
    th = threading.Thread(target=work, name="dsim-specialise-" + key, daemon=True)
    th.start()
    return _Job(thread=th)


def _spawn_process(t, key, cache_dir, log):
    """the compile in a DETACHED child process (`python -m diffrl_amd.specialise --ensure <template>`): it finishes -- lock, hipcc,
    rename into the cache -- whether or not this process is still there, so that a script shorter than the compile still leaves
    the library for its next run instead of starting the same compile again every time"""
    cdir = cache_dir or os.environ.get("DSIM_USER_LIBS") or USER_LIBS
    os.makedirs(cdir, exist_ok=True)
    tpath = os.path.join(cdir, "template_%s.npz" % key)
    tmp = tpath + ".tmp%d.npz" % os.getpid()
    t.save(tmp)
    os.replace(tmp, tpath)
    cmd = [sys.executable, "-m", "diffrl_amd.specialise", tpath, "--ensure", "--cache-dir", cdir]
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.dirname(HERE)] + [p for p in os.environ.get("PYTHONPATH", "").split(os.pathsep) if p]))
    return _Job(proc=subprocess.Popen(cmd, env=env, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, start_new_session=True))


_spawn_background = _spawn_process


def ensure_library_background(t, cache_dir=None, log=None):
    """The default of Engine(...) when hipcc is on PATH: the cached library of this model if there is one, else None -- and a
    background compile of it (ensure_library: same lock, same cache; a detached child process, _spawn_process), so that the NEXT
    Engine of this model picks it up while this one runs the generic kernels.  Returns (path or None, job or None); wait(t) joins
    a compile in flight."""
    path = cached_library(t, cache_dir)
    if path is not None or shutil.which("hipcc") is None:
        return path, None
    import hashlib
    key = hashlib.sha1(",".join(str(v) for v in flat_table(*layout(t))).encode()).hexdigest()[:12]
    job = _background.get(key)
    if job is None or not job.is_alive():
        job = _spawn_background(t, key, cache_dir, log)
        _background[key] = job
    return None, job


def wait(t=None, timeout=None):
    """joins the background compile of template t (all of them when t is None); True when none is left running"""
    import hashlib
    if t is None:
        jobs = list(_background.values())
    else:
        key = hashlib.sha1(",".join(str(v) for v in flat_table(*layout(t))).encode()).hexdigest()[:12]
        jobs = [_background[key]] if key in _background else []
    for j in jobs:
        j.join(timeout)
    return not any(j.is_alive() for j in jobs)


def ensure_library(t, cache_dir=None, log=None):
    """Path of a library that holds a specialised kernel set for template t, compiling one on first use
    (Engine(..., specialise=True) / DSIM_AUTO_SPECIALISE=1; the reference compiles ALL its kernels on first use,
    dflex/dflex/adjoint.py:2201-2263 -- here only a model that matches no compiled table does, and only when asked).

    * t matches a table of the product library: that library's path, nothing is built.
    * otherwise: <cache_dir>/libdsim_U<layout hash>_<source hash>.so; built with hipcc (generic kernels +
      this model's set, the recipe of `python -m diffrl_amd.specialise --only`) under a file lock, so that the ranks of a job
      that create the same model at the same time compile it once; written under a temporary name and renamed.
    * no hipcc, or the build fails (a tree shape one of the specialised phases refuses at compile time): a warning and None --
      the caller keeps the generic kernels, which run every model the layout builder accepts.
    cache_dir: default $DSIM_USER_LIBS, else csrc/user_libs/ (next to the product library, so that it travels with the package).
    The lock is flock(): keep the cache on a LOCAL filesystem (on NFS two ranks may both compile; the rename keeps the result
    whole either way).  Libraries of other source versions are left alone (two checkouts may share one cache) unless they are
    older than STALE_DAYS."""
    import fcntl
    import hashlib
    import warnings
    header_txt = open(HEADER).read()
    if os.environ.get("DSIM_LIB") is None and matches(t, header_txt):
        return capi.LIB_PATH
    flat = flat_table(*layout(t))
    name = "U" + hashlib.sha1(",".join(str(v) for v in flat).encode()).hexdigest()[:12]
    src = hashlib.sha1((source_hash() + "|" + " ".join(HIPCC_FLAGS)).encode()).hexdigest()[:10]
    cache_dir = cache_dir or os.environ.get("DSIM_USER_LIBS") or USER_LIBS
    out = os.path.join(cache_dir, "libdsim_%s_%s.so" % (name, src))
    if os.path.exists(out):
        return out
    if shutil.which("hipcc") is None:
        warnings.warn("diffrl_amd.specialise: hipcc not found, this model keeps the generic kernels")
        return None
    os.makedirs(cache_dir, exist_ok=True)
    say = log or (lambda m: print("[diffrl_amd.specialise] " + m, file=sys.stderr, flush=True))
    with open(os.path.join(cache_dir, ".lock_" + name), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.exists(out):   # another process built it while this one waited
                return out
            import time
            for f in os.listdir(cache_dir):   # half-written libraries of processes that exited while their background compile ran
                fp = os.path.join(cache_dir, f)
                if ".so.tmp" in f and time.time() - os.path.getmtime(fp) > 3600:
                    try:
                        os.remove(fp)
                    except OSError:
                        pass
            header = os.path.join(cache_dir, "dsim_static_layouts_%s.hpp" % name)
            open(header, "w").write(render([(name, t)]))
            tmp = out + ".tmp%d" % os.getpid()
            say("compiling a kernel set for this model (%d links, %d dofs, %d contacts): %s, about a minute, once"
                % (t.n_links, t.n_qd, t.n_contacts, out))
            try:
                build_library(tmp, header, [name], quiet=True)
            except subprocess.CalledProcessError as e:
                if os.path.exists(tmp):
                    os.remove(tmp)
                warnings.warn("diffrl_amd.specialise: hipcc refused the specialised kernels of this model (it keeps the generic "
                              "ones):\n%s" % (e.output or b"").decode(errors="replace")[-2000:])
                return None
            os.replace(tmp, out)
            import time
            for f in os.listdir(cache_dir):   # this model's libraries of other source versions, once nobody has used them for weeks
                fp = os.path.join(cache_dir, f)
                if (f.startswith("libdsim_%s_" % name) and f.endswith(".so") and fp != out
                        and time.time() - max(os.path.getmtime(fp), os.path.getatime(fp)) > STALE_DAYS * 86400):
                    try:
                        os.remove(fp)
                    except OSError:
                        pass
            return out
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m diffrl_amd.specialise", description=__doc__.split("\n\n")[0])
    ap.add_argument("template", help=".npz written by ArticulationTemplate.save()")
    ap.add_argument("--name", help="name of the kernel set (letters / digits)")
    ap.add_argument("--ensure", action="store_true", help="(what the default Engine(...) runs in the background) compile this model's "
                                                          "kernel set into the cache unless it is there already, and print its path")
    ap.add_argument("--cache-dir", help="(with --ensure) the cache directory (default $DSIM_USER_LIBS, else csrc/user_libs/)")
    ap.add_argument("--header-out", help="write the generated header here instead of csrc/dsim_static_layouts.hpp (the in-tree header "
                                         "and csrc/user_models/ stay untouched)")
    ap.add_argument("--lib-out", help="build this library instead of csrc/libdsim_hip.so (use it with DSIM_LIB=<path>)")
    ap.add_argument("--only", action="store_true", help="compile the generic kernels + this model's set only (~40 s instead of minutes)")
    ap.add_argument("--no-build", action="store_true", help="generate the header, do not run hipcc")
    ap.add_argument("--also", action="append", default=[], metavar="NAME=template.npz",
                    help="(with --header-out) further user models for the same header / the same --only library; may be repeated")
    a = ap.parse_args(argv)
    t = ArticulationTemplate.load(a.template)
    if a.ensure:
        path = ensure_library(t, a.cache_dir)
        print(path or "no library (hipcc missing or the build failed): the generic kernels stay")
        return 0 if path else 1
    if not a.name:
        ap.error("--name is required (or --ensure)")
    models = shipped_templates()
    # everything that can refuse the request is checked BEFORE anything is written: a bad name must not leave a template in
    # csrc/user_models/ that breaks every later regeneration (tools/gen_static_layouts.py included)
    if not re.fullmatch(r"[A-Za-z][A-Za-z0-9]*", a.name):
        ap.error("--name %r: letters and digits only, starting with a letter (it becomes part of C++ identifiers)" % a.name)
    if a.name in [tag for tag, _ in models]:
        ap.error("--name %s is the name of a shipped kernel set; choose another" % a.name)
    if a.only and not a.lib_out and not a.no_build:
        ap.error("--only builds a library WITHOUT the shipped specialised sets (Ant, Humanoid, ... would silently fall back to the "
                 "generic kernels at about half speed): it needs --lib-out <path> so that csrc/libdsim_hip.so is not overwritten")
    known = matches(t, render(models))
    if known:
        print("this model already has a specialised kernel set: %s (nothing to do)" % known)
        return 0
    extra = []
    for spec in a.also:
        n2, _, path2 = spec.partition("=")
        if not re.fullmatch(r"[A-Za-z][A-Za-z0-9]*", n2) or not path2 or n2 == a.name or n2 in [tag for tag, _ in models]:
            ap.error("--also %r: expected NAME=template.npz with a fresh identifier as NAME" % spec)
        if not a.header_out:
            ap.error("--also belongs to --header-out builds (the in-tree header takes its user models from csrc/user_models/)")
        extra.append((n2, ArticulationTemplate.load(path2)))
    if a.header_out:
        models = models + [(n, u) for n, u in user_templates() if n != a.name and n not in [e[0] for e in extra]] + [(a.name, t)] + extra
        header = a.header_out
    else:
        models = models + [(n, u) for n, u in user_templates() if n != a.name] + [(a.name, t)]
        header = HEADER
    txt = render(models)   # (raises on anything the layout builder refuses: still nothing written)
    if not a.header_out:
        os.makedirs(USER_DIR, exist_ok=True)
        t.save(os.path.join(USER_DIR, a.name + ".npz"))
    open(header, "w").write(txt)
    assert matches(t, txt) == a.name
    off, dims = layout(t)
    print("wrote %s: %d models, %s = %d links, %d dofs, %d contacts, LDS image %d / %d words (forward / adjoint)"
          % (header, len(models), a.name, dims["L"], dims["nd"], dims["C"], off["fwd_words"], off["total_words"]))
    if a.no_build:
        return 0
    out = a.lib_out or os.path.join(CSRC, "libdsim_hip.so")
    build_library(out, header if a.header_out else None, ([a.name] + [e[0] for e in extra]) if a.only else None)
    print("built %s; a model created from this template now reports dsim_model_variant > 0%s"
          % (out, "" if not a.lib_out else " (run with DSIM_LIB=%s)" % os.path.abspath(out)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
