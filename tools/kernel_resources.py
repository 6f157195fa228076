#!/usr/bin/env python3
"""Per-kernel resources of the built library from its code-object metadata (no GPU): VGPR / AGPR / SGPR per wave, static
LDS bytes, scratch bytes per lane, spilled VGPRs.   usage: kernel_resources.py [lib.so] > profiles/rNN_kernel_resources.txt"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "dash-infer_amd", "lib", "libdashinfer_hip.so")
    tmp = tempfile.mkdtemp(prefix="dihip_res_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        rows = []
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", os.path.join(tmp, f)], stdout=subprocess.PIPE, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                def field(k, blk=blk):
                    m = re.search(r"\." + k + r":\s+(\S+)", blk)
                    return m.group(1) if m else "0"
                name = field("name")
                demangled = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
                demangled = re.sub(r"\(.*\)$", "", demangled)
                agpr = re.match(r"\s*(\d+)", blk).group(1)
                rows.append((demangled, int(field("vgpr_count")), int(agpr), int(field("sgpr_count")), int(field("group_segment_fixed_size")),
                             int(field("private_segment_fixed_size")), int(field("vgpr_spill_count"))))
        print("# hipcc --offload-arch=gfx950 -O3 (the library's flags): per-kernel resources from the code-object metadata")
        print("# VGPR / AGPR / SGPR per wave, static LDS bytes, scratch bytes per lane, spilled VGPRs; %d kernels, %d with spills"
              % (len(rows), sum(1 for r in rows if r[6])))
        for r in sorted(set(rows)):
            print("%-110s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %4d spill %3d" % ((r[0][:110],) + r[1:]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
