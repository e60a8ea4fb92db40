"""Developer tool / test helper: per-kernel code-object metadata of a built libdsim_hip.so (no GPU needed): VGPRs, SGPR
spills, scratch bytes, LDS.  usage: python tools/kernel_meta.py [lib.so] [substring]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--unbundle", "--input=" + fat, "--output=" + co], stderr=subprocess.DEVNULL)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    out, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s+- \.agpr_count:\s+(\d+)", line) or re.match(r"\s+\.agpr_count:\s+(\d+)", line)
        if m:
            cur = {"agpr_count": int(m.group(1))}
            out.append(cur)
            continue
        m = re.match(r"\s+\.(\w+):\s+(\S+)", line)
        if m and cur is not None and m.group(1) in ("name", "vgpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count",
                                                     "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
            v = m.group(2)
            cur[m.group(1)] = int(v) if v.isdigit() else v
    return [k for k in out if "name" in k]


def short(name):
    m = re.search(r"(dsim_\w+_kernel)I\d+DsimOff(\w*?)\d+DsimDims\w*?Li(\d)(?:ELb(\d))?(?:EL[bi](\d))?", name)
    if not m:
        return name[:50]
    return "%s<%s, waves %s%s%s>" % (m.group(1), m.group(2) or "generic", m.group(3), ", lean" if m.group(4) == "1" else "",
                                     {"1": ", helper", "2": ", pair"}.get(m.group(5), ""))


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = (sys.argv[1] if len(sys.argv) > 1 else "") or os.path.join(root, "diffrl_amd", "csrc", "libdsim_hip.so")
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    print("%-52s %5s %5s %7s %7s %5s" % ("kernel", "VGPR", "AGPR", "sgprSp", "scratch", "wg"))
    for k in kernels(lib):
        if pat in k["name"]:
            print("%-52s %5s %5s %7s %7s %5s" % (short(k["name"]), k.get("vgpr_count"), k.get("agpr_count"), k.get("sgpr_spill_count"),
                                                  k.get("private_segment_fixed_size"), k.get("max_flat_workgroup_size")))
