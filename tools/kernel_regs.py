#!/usr/bin/env python3
"""VGPR / AGPR / spill / LDS / scratch of the kernels in libwtpse_hip.so whose (demangled-ish) name matches a regex (llvm-readelf notes).
    python tools/kernel_regs.py 'conv_x3_kILi3ELi1E'   """
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "wt-pse-code_amd", "wtpse_hip", "libwtpse_hip.so")
BIN = "/opt/rocm/lib/llvm/bin"
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
tmp = tempfile.mkdtemp(prefix="regs_")
try:
    shutil.copy(LIB, os.path.join(tmp, "lib.so"))
    subprocess.run([os.path.join(BIN, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
    rows = []
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" not in f:
            continue
        txt = subprocess.run([os.path.join(BIN, "llvm-readelf"), "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
        for e in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, e).group(1)) if re.search(r"\.%s:\s+(\d+)" % k, e) else 0
            name = re.search(r"\.name:\s+(\S+)", e).group(1)
            if pat.search(name):
                rows.append((name, g("vgpr_count"), int(re.match(r"\s*(\d+)", e).group(1)), g("vgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
    for r in sorted(rows):
        print("%-90s vgpr %3d agpr %3d spill %3d lds %6d scratch %4d" % r)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
