#!/usr/bin/env python3
"""Register / LDS / spill summary of the kernels in a gfx950 object or .so (reads the code-object
metadata note through llvm-readelf).  Usage: python tools/kres.py egt_amd/lib/libegt_amd.so [name-filter]"""
import re
import subprocess
import sys

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CLANG_OFFLOAD = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"


def notes(path):
    out = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
    if ".name:" in out:
        return out
    # host object / fat binary: pull the device code object out first
    import tempfile, os
    tmp = tempfile.mkdtemp()
    dev = os.path.join(tmp, "dev.co")
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], capture_output=True)
    subprocess.run([CLANG_OFFLOAD, "--unbundle", "--type=o", f"--input={fat}", f"--output={dev}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True)
    if os.path.exists(dev):
        return subprocess.run([READELF, "--notes", dev], capture_output=True, text=True).stdout
    return out


def main():
    txt = notes(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        if flt and flt not in name:
            continue
        agpr = blk.strip().split()[0]
        print(f"{name[:110]:110s} vgpr {g('vgpr_count'):>4s} agpr {agpr:>4s} sgpr {g('sgpr_count'):>4s} "
              f"spill {g('vgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")


if __name__ == "__main__":
    main()
