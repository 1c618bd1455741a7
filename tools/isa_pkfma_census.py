#!/usr/bin/env python3
"""Census of the packed fp32 FMAs with operand-select modifiers in libwtpse_hip.so (llvm-objdump, no GPU): per kernel, how many
`v_pk_fma_f32` BROADCAST an operand out of a register pair (op_sel[i] == op_sel_hi[i]: both result halves read the same half of
source i), how many operands of one instruction are broadcast, and whether the broadcast sources are vector or scalar registers.
Round 3's mask corruption came from an instruction with TWO VGPR-pair broadcasts (`op_sel_hi:[1,0,0]` / `op_sel:[0,1,1]`: scale and
shift of the ReLU decision); tests/test_isa_checks.py fails the build on that form in ANY kernel.

    python tools/isa_pkfma_census.py [path/to/libwtpse_hip.so]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="wtpse_isa_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([OBJDUMP, "--offloading", so], cwd=tmp, check=True, capture_output=True)
        kernels = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                    kernels[cur] = []
                elif cur is not None and line.startswith("\t"):
                    kernels[cur].append(line.strip())
        return kernels
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def classify(line):
    """-> list of (source index, 'v' | 's' | other) of the broadcast operands of one v_pk_fma_f32 line."""
    m = re.match(r"v_pk_fma_f32\s+(\S+),\s*(\S+),\s*(\S+),\s*(\S+?)(\s|$)", line)
    if not m:
        return None
    srcs = [m.group(i).rstrip(",") for i in (2, 3, 4)]
    sel = [0, 0, 0]
    hi = [1, 1, 1]
    ms = re.search(r"op_sel:\[([01]),([01]),([01])\]", line)
    mh = re.search(r"op_sel_hi:\[([01]),([01]),([01])\]", line)
    if ms:
        sel = [int(x) for x in ms.groups()]
    if mh:
        hi = [int(x) for x in mh.groups()]
    out = []
    for i in range(3):
        if sel[i] == hi[i]:
            kind = "v" if srcs[i].startswith(("v[", "v", "a[")) and not srcs[i].startswith("vcc") else ("s" if srcs[i].startswith("s") else "c")
            out.append((i, kind))
    return out


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return dict(zip(names, out.splitlines()))
    except Exception:
        return {n: n for n in names}


def census(kernels):
    rows = {}
    for name, body in kernels.items():
        c = Counter()
        for l in body:
            if not l.startswith("v_pk_fma_f32"):
                continue
            c["pk_fma"] += 1
            b = classify(l)
            if b is None or not b:
                continue
            nv = sum(1 for _, k in b if k == "v")
            c["bcast"] += 1
            c["bcast_vgpr_%d" % nv] += 1
            if nv == 0:
                c["bcast_sgpr_only"] += 1
        if c["pk_fma"]:
            rows[name] = c
    return rows


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "wt-pse-code_amd", "wtpse_hip", "libwtpse_hip.so")
    rows = census(disassemble(lib))
    dm = demangle(list(rows))
    tot = Counter()
    fam = defaultdict(Counter)
    for n, c in rows.items():
        tot.update(c)
        fam[re.sub(r"<.*", "", dm[n].replace("void ", ""))].update(c)
    print("kernels with v_pk_fma_f32: %d; instructions %d; with a broadcast operand %d (SGPR-only %d, one VGPR-pair broadcast %d, "
          "TWO VGPR-pair broadcasts %d, three %d)" % (len(rows), tot["pk_fma"], tot["bcast"], tot["bcast_sgpr_only"], tot["bcast_vgpr_1"],
                                                      tot["bcast_vgpr_2"], tot["bcast_vgpr_3"]))
    print("%-28s %8s %8s %10s %10s %10s" % ("kernel family", "pk_fma", "bcast", "sgpr-only", "1 vgpr", ">=2 vgpr"))
    for f, c in sorted(fam.items(), key=lambda kv: -kv[1]["pk_fma"]):
        print("%-28s %8d %8d %10d %10d %10d" % (f[:28], c["pk_fma"], c["bcast"], c["bcast_sgpr_only"], c["bcast_vgpr_1"],
                                                 c["bcast_vgpr_2"] + c["bcast_vgpr_3"]))


if __name__ == "__main__":
    main()
