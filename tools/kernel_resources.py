#!/usr/bin/env python3
"""Register / LDS / scratch budget of the kernels in the SHIPPED library (or one object file), read from the code objects' metadata
notes -- no GPU needed.   python tools/kernel_resources.py [pattern] [--so PATH]
Columns: VGPRs (arch), AGPRs, SGPRs, LDS bytes, private segment (scratch = spill) bytes, spilled VGPRs, waves per SIMD the VGPR
count allows (512 / ceil8(vgpr + agpr))."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_count  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pattern", nargs="?", default="msm_accumulate29_kernel")
    ap.add_argument("--so", default=os.path.join(ROOT, "gnark_amd", "libgnark_amd.so"))
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as wd:
        for co in isa_count.code_objects(a.so, wd):
            txt = subprocess.run([os.path.join(isa_count.LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", "\n" + txt)[1:]:
                blk = ".agpr_count:" + blk
                f = {k: v for k, v in re.findall(r"\.(\w+):\s+(\S+)", blk)}
                name = f.get("name", "")
                dem = isa_count.demangle([name])[name]
                if a.pattern not in dem:
                    continue
                v, ag = int(f.get("vgpr_count", 0)), int(f.get("agpr_count", 0))
                tot = (v + ag + 7) // 8 * 8
                print("%-110s vgpr %3d agpr %3d sgpr %3s lds %6s scratch %5s spilled_vgprs %3s waves/SIMD<=%d" % (
                    dem.split("(")[0][:110], v, ag, f.get("sgpr_count"), f.get("group_segment_fixed_size"), f.get("private_segment_fixed_size"),
                    f.get("vgpr_spill_count"), min(8, 512 // max(tot, 1))))


if __name__ == "__main__":
    main()
