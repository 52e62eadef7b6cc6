"""Per-kernel register / scratch / LDS figures of the built library (read from the code object's metadata notes; no GPU needed).
    python tools/kernel_resources.py [substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "myosuite_mjx_amd", "libmyo_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def resources(lib=LIB):
    tmp = "/tmp/_myo_co"
    os.makedirs(tmp, exist_ok=True)
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={lib}", f"--output={tmp}/k.co"], stderr=subprocess.DEVNULL) if False else None
    # the fat binary sits in .hip_fatbin; extract it, then unbundle
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, f"{tmp}/fat.bin"])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={tmp}/fat.bin", f"--output={tmp}/k.co"])
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", f"{tmp}/k.co"], text=True)
    out = []
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
        name = g("name").group(1)
        out.append(dict(name=name, vgpr=int(g("vgpr_count").group(1)), agpr=int(blk.split()[0]), sgpr=int(g("sgpr_count").group(1)),
                        vgpr_spill=int(g("vgpr_spill_count").group(1)), sgpr_spill=int(g("sgpr_spill_count").group(1)),
                        scratch=int(g("private_segment_fixed_size").group(1)), lds=int(g("group_segment_fixed_size").group(1))))
    return out


if __name__ == "__main__":
    sub = sys.argv[1] if len(sys.argv) > 1 else ""
    for r in resources():
        d = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        if sub in d:
            print(f"{d[:110]:110s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} spill v{r['vgpr_spill']:3d} s{r['sgpr_spill']:3d} scratch {r['scratch']:4d} B")
