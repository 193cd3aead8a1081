#!/usr/bin/env python3
"""VGPRs, scratch bytes and spills of every kernel in the built library (read from the code object's notes).
usage: python tools/kernel_regs.py [lib.so] [substring ...]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    args = sys.argv[1:]
    so = args.pop(0) if args and args[0].endswith(".so") else os.path.join(ROOT, "grasptrajopt_amd", "csrc", "libgto_hip.so")
    with tempfile.TemporaryDirectory() as d:
        fb, co = os.path.join(d, "fb"), os.path.join(d, "co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb, so], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fb,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
        names = subprocess.run(["c++filt"], input=notes, capture_output=True, text=True).stdout
    cur = {}
    for line in names.splitlines():
        line = line.strip().lstrip("- ")
        for k in (".name:", ".vgpr_count:", ".private_segment_fixed_size:", ".group_segment_fixed_size:", ".vgpr_spill_count:"):
            if line.startswith(k):
                cur[k] = line[len(k):].strip()
        if line.startswith(".vgpr_spill_count:"):
            n = cur.get(".name:", "?").split("(")[0]
            if not args or any(a in n for a in args):
                print(f"{n:44s} vgpr {cur.get('.vgpr_count:'):>4s}  scratch {cur.get('.private_segment_fixed_size:'):>5s} B  "
                      f"spills {cur.get('.vgpr_spill_count:'):>4s}  static lds {cur.get('.group_segment_fixed_size:'):>6s} B")
            cur = {}


if __name__ == "__main__":
    main()
