#!/usr/bin/env python3
"""Resource usage of every gfx950 kernel of libdotsocp, from the compiler's own per-kernel summary: each .hip file is
compiled to device assembly with the flags of csrc/Makefile (`hipcc -S --offload-device-only`), and the
`; Kernel info:` block that the AMDGPU backend appends to every kernel is tabulated -- VGPRs, AGPRs, SGPRs, scratch
(spill) bytes, static LDS bytes, occupancy in waves per SIMD.  Runs without a GPU.

    python tools/isa_stats.py > profiles/r02_isa_stats.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dot-socp_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include")]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines()


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
            asm = os.path.join(tmp, src[:-4] + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--offload-device-only", os.path.join(CSRC, src), "-o", asm],
                           check=True, stderr=subprocess.DEVNULL)
            text = open(asm).read()
            # "\t.globl\t<name>" ... "; Kernel info:" blocks follow each kernel's code
            for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)^; Kernel info:\n(.*?)^; WaveLimiterHint", text, re.S | re.M):
                name, info = m.group(1), m.group(3)
                get = lambda k: int(re.search(r"; %s: (\d+)" % k, info).group(1))      # noqa: E731
                rows.append((src, name, get("NumVgprs"), get("NumAgprs"), get("TotalNumSgprs"), get("ScratchSize"),
                             get("LDSByteSize"), get("Occupancy")))
    names = demangle([r[1] for r in rows])
    print("# gfx950 kernel resources of libdotsocp (compiler summary; flags: %s)" % " ".join(FLAGS[:-1]))
    print("# occupancy = waves per SIMD the register / LDS budget allows; scratch = spill bytes per lane (0 is the bar)")
    print("%-16s %5s %5s %5s %8s %8s %4s  %s" % ("file", "VGPR", "AGPR", "SGPR", "scratch", "LDS", "occ", "kernel"))
    spills = 0
    for (src, _, v, a, sg, sc, lds, occ), nm in zip(rows, names):
        nm = re.sub(r"^void ", "", nm)
        nm = re.sub(r"\(.*", "", nm).replace("dotsocp::", "")
        print("%-16s %5d %5d %5d %8d %8d %4d  %s" % (src, v, a, sg, sc, lds, occ, nm))
        spills += sc > 0
    print("# %d kernels, %d with scratch" % (len(rows), spills))
    return 0


if __name__ == "__main__":
    sys.exit(main())
