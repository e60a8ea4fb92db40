"""Static opcode census of the built kernels (no GPU needed): which share of a kernel's VALU instructions carries floating-point
work, and how many flops each carries -- so that bench.py can turn the measured VALU instruction count (SQ_INSTS_VALU) and
active lanes (SQ_THREAD_CYCLES_VALU) into a flop rate instead of counting every VALU instruction as an FMA.

    python tools/opcode_census.py [lib.so] [--out profiles/opcode_census.json]

Disassembles the gfx950 code object of the library (llvm-objdump), and for every env-step kernel takes the instructions of its
substep loop (the longest backward branch of the kernel: the loop over the substeps holds > 95 % of the dynamic instruction
count -- 16 to 48 iterations against a prologue / epilogue executed once) and classifies them:

    fma     v_fma / v_fmac / v_mac / v_mad / v_pk_fma (f32)      2 flop      (packed: x2)
    arith   v_mul / v_add / v_sub / v_pk_mul / v_pk_add (f32)    1 flop
    special v_rcp / v_rsq / v_sqrt / v_sin / v_cos / v_exp ...   1 flop
    minmax  v_min / v_max / v_med3 (f32)                         1 flop
    other   moves, selects, compares, integer / address arithmetic, conversions, cross-lane moves: 0 flop

The census is STATIC (one pass over the loop body, both sides of every branch counted once): the refresh path of the mass
matrix, executed once per group of substeps, is over-represented; it is multiply-add dense, so the flop share is, if anything,
over-stated.  Written next to the profiles with the hash of the kernel sources; bench.py quotes it only for matching sources.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FMA = re.compile(r"^v_(fma|fmac|mac|mad|fmaak|fmamk|madak|madmk)_(legacy_)?f32|^v_pk_fma_f32")
ARITH = re.compile(r"^v_(mul|add|sub|subrev)_(legacy_)?f32|^v_pk_(mul|add)_f32|^v_mul_legacy_f32")
SPECIAL = re.compile(r"^v_(rcp|rsq|sqrt|sin|cos|exp|log|rcp_iflag|frexp_mant|ldexp|fract|floor|ceil|rndne|trunc)_(legacy_)?f32|^v_div_(scale|fmas|fixup)_f32")
MINMAX = re.compile(r"^v_(min|max|med3|min3|max3)_f32")


def disassemble(lib):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--unbundle", "--input=" + fat, "--output=" + co], stderr=subprocess.DEVNULL)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout


def kernels(dis):
    """{symbol: [(address, opcode, branch target or None)]}"""
    out, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\s.*//\s*([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            tgt = None
            if m.group(1).startswith("s_cbranch") or m.group(1) == "s_branch":
                t = re.search(r"<\S+\+0x([0-9a-f]+)>", line)
                tgt = int(t.group(1), 16) if t else None
            cur.append((int(m.group(2), 16), m.group(1), tgt))
    return out


def short(name):
    m = re.search(r"(dsim_env_(?:fwd|bwd)_kernel)I\d+DsimOff(\w*?)\d+DsimDims\w*?Li(\d)ELb(\d)ELi(\d)", name)
    if not m or m.group(4) == "1":
        return None   # not an env-step kernel, or a lean-checkpoint instantiation
    return "%s<%s,%s>" % (m.group(1), m.group(2) or "generic", {"0": "plain", "1": "helper", "2": "pair"}[m.group(5)])


def census(insts):
    base = insts[0][0]
    best = None
    for i, (addr, op, tgt) in enumerate(insts):
        if tgt is not None and base + tgt <= addr:   # backward branch: a loop
            j = next((k for k, x in enumerate(insts) if x[0] >= base + tgt), None)
            if j is not None and (best is None or i - j > best[1] - best[0]):
                best = (j, i)
    body = insts[best[0]:best[1] + 1] if best else insts
    c = collections.Counter()
    for _, op, _ in body:
        o = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
        if not o.startswith("v_") or o.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            c["non_valu"] += 1
            continue
        c["valu"] += 1
        pk = 2 if o.startswith("v_pk_") else 1
        if FMA.match(o):
            c["fma"] += 1
            c["flops"] += 2 * pk
        elif ARITH.match(o) or SPECIAL.match(o) or MINMAX.match(o):
            c["arith"] += 1
            c["flops"] += pk
        else:
            c["other"] += 1
    return {"loop_insts": len(body), "valu": c["valu"], "fma": c["fma"], "arith_special_minmax": c["arith"], "non_flop_valu": c["other"],
            "flops_per_valu_inst": c["flops"] / max(c["valu"], 1), "flop_carrying_share": (c["fma"] + c["arith"]) / max(c["valu"], 1)}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(ROOT, "diffrl_amd", "csrc", "libdsim_hip.so")
    out = os.path.join(ROOT, "profiles", "opcode_census.json")
    if "--out" in sys.argv:
        out = sys.argv[sys.argv.index("--out") + 1]
    sys.path.insert(0, ROOT)
    import bench
    rec = {"csrc_hash": bench.csrc_hash(), "library": os.path.relpath(lib, ROOT), "method": __doc__.split("\n\n")[2], "kernels": {}}
    for sym, insts in kernels(disassemble(lib)).items():
        k = short(sym)
        if k and insts:
            rec["kernels"][k] = census(insts)
    json.dump(rec, open(out, "w"), indent=1)
    for k, v in sorted(rec["kernels"].items()):
        print("%-46s loop %5d  valu %5d  fma %4d  arith %4d  other %4d  flop/valu %.3f" %
              (k, v["loop_insts"], v["valu"], v["fma"], v["arith_special_minmax"], v["non_flop_valu"], v["flops_per_valu_inst"]))
    print("wrote", out)


if __name__ == "__main__":
    main()
