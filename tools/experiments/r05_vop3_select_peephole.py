#!/usr/bin/env python3
"""Builds libokrt.so through an assembly peephole pass (experiment / optional build step).

  python3 tools/build_peephole.py [--out gpu-raytracer_amd/libokrt.so] [--files okrt_fast_trace.hip,...] [--dry]

Why: tools/calib/issue_rate.hip measures `v_cndmask_b32_e32 vD, a, vB, vcc` (the VOP2 encoding with its implicit VCC operand) at
one instruction per ~23 shader clocks on MI355X, and the SAME select in the VOP3 encoding (`v_cndmask_b32_e64 vD, a, vB, vcc`) at one
per 4 (profiles/r05_issue_rate.txt). The compiler shrinks every select it can to VOP2. This script compiles each csrc/*.hip to
gfx950 assembly, rewrites those selects to their VOP3 form (same operands, same semantics; only operands VOP3 can encode on gfx9:
registers and inline constants, never a 32-bit literal), assembles, links and bundles the code object the way hipcc does, and compiles
the host side against it. Everything else in the build is gpu-raytracer_amd/build.py's.
"""
import argparse
import glob
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "gpu-raytracer_amd")
LLVM = "/opt/rocm/lib/llvm/bin"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value"]

INLINE = re.compile(r"^(-?\d+|-?0\.5|-?1\.0|-?2\.0|-?4\.0|v\d+|s\d+|vcc_lo|vcc_hi|-v\d+|\|v\d+\|)$")
SEL = re.compile(r"^(\s*)v_cndmask_b32_e32 (v\d+), ([^,]+), (v\d+), vcc\s*$")


def peephole(text):
    out, n, kept = [], 0, 0
    for line in text.split("\n"):
        m = SEL.match(line)
        if m:
            src0 = m.group(3).strip()
            ok = False
            if INLINE.match(src0):
                ok = not src0.lstrip("-").isdigit() or -16 <= int(src0) <= 64
            if ok:
                out.append(f"{m.group(1)}v_cndmask_b32_e64 {m.group(2)}, {src0}, {m.group(4)}, vcc")
                n += 1
                continue
            kept += 1
        out.append(line)
    return "\n".join(out), n, kept


def run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(PKG, "libokrt.so"))
    ap.add_argument("--files", default="", help="comma-separated csrc/*.hip to rewrite (default: all)")
    ap.add_argument("--dry", action="store_true", help="only count the selects that would be rewritten")
    args = ap.parse_args()
    objdir = os.path.join(PKG, "_build", "peephole")
    os.makedirs(objdir, exist_ok=True)
    src = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.cpp")))
    chosen = set(args.files.split(",")) if args.files else None
    objs = []
    for f in src:
        base = os.path.basename(f)
        obj = os.path.join(objdir, base + ".o")
        objs.append(obj)
        if not f.endswith(".hip") or (chosen is not None and base not in chosen):
            if not args.dry:
                run([HIPCC] + FLAGS + ["-c", f, "-o", obj])
            continue
        s = os.path.join(objdir, base + ".s")
        run([HIPCC] + FLAGS + ["--cuda-device-only", "-S", f, "-o", s])
        text, n, kept = peephole(open(s).read())
        print(f"[peephole] {base}: {n} selects VOP2 -> VOP3, {kept} left (literal operand)")
        if args.dry:
            continue
        open(s, "w").write(text)
        dev_o, hsaco, fb = s + ".dev.o", s + ".hsaco", s + ".hipfb"
        run([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", dev_o])
        run([os.path.join(LLVM, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, dev_o])
        run([os.path.join(LLVM, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
             "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + hsaco, "-output=" + fb])
        run([HIPCC] + FLAGS + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", f, "-o", obj])
    if not args.dry:
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", args.out] + objs + ["-ldl"])
        print("[peephole] wrote", args.out)


if __name__ == "__main__":
    main()
