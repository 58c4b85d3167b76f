#!/usr/bin/env python3
"""Instruction mix of the hot loops, read from the SHIPPED library (gnark_amd/libgnark_amd.so), no GPU needed.

  python tools/isa_count.py                       # the default kernel list (bucket accumulation G1/G2, window reduction, NTT pass)
  python tools/isa_count.py msm_accumulate29_kernel --dump /tmp/acc.s

How: the gfx950 code objects are cut out of the .hip_fatbin section (clang offload bundles), disassembled with
llvm-objdump, and for every kernel whose demangled name contains the pattern the LARGEST loop (a backward s_cbranch / s_branch
whose body holds the most v_mad_u64_u32) is taken as "the" loop: for msm_accumulate29_kernel that is one mixed addition per
iteration, for ntt_pass29r4_kernel one radix-4 block (4 butterflies) per iteration of the stage loop.  Instructions are
classed with the issue rates measured by ga_microbench (profiles/README.md): v_mad_u64_u32 (5.1 cycles per wave), other
half-rate VALU (64-bit shifts/adds, 32-bit multiplies; 4.6), full-rate VALU (2.3), LDS, VMEM, scalar.
This is the static count of the loop body: conditional side paths inside the body (e.g. the first point of a task) are
included, so it is an upper bound of the per-iteration count by a few dozen instructions.
"""
import argparse
import collections
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
HALF_RATE = ("v_lshl_add_u64", "v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32",
             "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_u32_u24", "v_mad_i64_i32", "v_add_f64", "v_mul_f64", "v_fma_f64")
DEFAULT = ["msm_accumulate29_kernel", "msm_reduce_groups29_kernel", "ntt_pass29r4_kernel"]


def code_objects(so_path, workdir):
    fat = os.path.join(workdir, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path, os.path.join(workdir, "unused.so")])
    d = open(fat, "rb").read()
    out = []
    i = d.find(MAGIC)
    while i >= 0:
        nb = struct.unpack_from("<Q", d, i + 24)[0]
        p = i + 32
        for _ in range(nb):
            off, size, idlen = struct.unpack_from("<QQQ", d, p)
            p += 24
            tid = d[p:p + idlen].decode()
            p += idlen
            if "gfx950" in tid and size:
                path = os.path.join(workdir, "co%d.elf" % len(out))
                open(path, "wb").write(d[i + off:i + off + size])
                out.append(path)
        i = d.find(MAGIC, i + 1)
    return out


def disassemble(path):
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", "--no-show-raw-insn", path], capture_output=True, text=True,
                         check=True).stdout
    funcs, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m:
            funcs[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return funcs


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, r.stdout.splitlines()))


def classify(mn):
    if mn == "v_mad_u64_u32":
        return "v_mad_u64_u32"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mn.startswith("s_"):
        return "scalar"
    if mn.startswith("v_"):
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)
        return "valu_half_rate" if base in HALF_RATE else "valu_full_rate"
    return "other"


def loops(insns):
    """[(start_index, end_index)] of backward branches (the loop body is insns[start..end])"""
    addr_to_idx = {a: k for k, (a, _, _) in enumerate(insns)}
    out = []
    for k, (a, mn, ops) in enumerate(insns):
        if mn.startswith(("s_cbranch", "s_branch")):
            m = re.search(r"(-?\d+)\s*$", ops)
            if not m:
                continue
            simm = int(m.group(1))
            if simm >= 32768:
                simm -= 65536
            target = a + 4 + 4 * simm
            if target <= a and target in addr_to_idx:
                out.append((addr_to_idx[target], k))
    return out


def analyse(name, insns):
    best = None
    for lo, hi in loops(insns):
        mads = sum(1 for _, mn, _ in insns[lo:hi + 1] if mn == "v_mad_u64_u32")
        if best is None or mads > best[0]:
            best = (mads, lo, hi)
    if best is None:
        return None
    _, lo, hi = best
    body = insns[lo:hi + 1]
    classes = collections.Counter(classify(mn) for _, mn, _ in body)
    top = collections.Counter(re.sub(r"_(e32|e64)$", "", mn) for _, mn, _ in body)
    cyc = classes["v_mad_u64_u32"] * 5.1 + classes["valu_half_rate"] * 4.6 + classes["valu_full_rate"] * 2.3
    return {"kernel": name, "loop_instructions": len(body), "classes": dict(classes), "issue_cycles_per_wave_iteration": round(cyc),
            "top": top.most_common(14), "whole_kernel_instructions": len(insns), "range": (lo, hi)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("patterns", nargs="*", default=DEFAULT)
    ap.add_argument("--so", default=os.path.join(ROOT, "gnark_amd", "libgnark_amd.so"))
    ap.add_argument("--dump", help="write the disassembly of the selected loop of the first matching kernel here")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as wd:
        results = []
        for co in code_objects(args.so, wd):
            funcs = disassemble(co)
            names = demangle(list(funcs))
            for sym, insns in funcs.items():
                dn = names.get(sym, sym)
                if sym.endswith(".kd") or not any(p in dn for p in args.patterns) or "redo" in dn:
                    continue
                r = analyse(dn, insns)
                if r:
                    results.append(r)
                    if args.dump:
                        lo, hi = r["range"]
                        with open(args.dump, "w") as f:
                            f.write("// %s\n" % dn)
                            for a, mn, ops in insns[lo:hi + 1]:
                                f.write("%08x  %s %s\n" % (a, mn, ops))
                        args.dump = None
        for r in sorted(results, key=lambda r: r["kernel"]):
            c = r["classes"]
            print("%s\n  loop: %d instructions  (v_mad_u64_u32 %d, other half-rate VALU %d, full-rate VALU %d, LDS %d, VMEM %d, scalar %d)  "
                  "=> %d issue cycles per wave-iteration" % (r["kernel"][:150], r["loop_instructions"], c.get("v_mad_u64_u32", 0),
                                                             c.get("valu_half_rate", 0), c.get("valu_full_rate", 0), c.get("lds", 0),
                                                             c.get("vmem", 0), c.get("scalar", 0), r["issue_cycles_per_wave_iteration"]))
            print("  top:", ", ".join("%s %d" % kv for kv in r["top"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
