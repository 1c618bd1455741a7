#!/usr/bin/env python3
"""Disassembly of one kernel of libwtpse_hip.so (first symbol matching the regex) + a static instruction census per basic block.
    python tools/kernel_isa.py 'conv_x3r_kILi2ELi1ELi4ELi5ELi0ELi2ELb0' [out.s]"""
import os, re, shutil, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "wt-pse-code_amd", "wtpse_hip", "libwtpse_hip.so")
BIN = "/opt/rocm/lib/llvm/bin"
pat = re.compile(sys.argv[1])
tmp = tempfile.mkdtemp(prefix="isa_")
try:
    shutil.copy(LIB, os.path.join(tmp, "lib.so"))
    subprocess.run([os.path.join(BIN, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
    for f in sorted(os.listdir(tmp)):
        if "amdgcn" not in f:
            continue
        txt = subprocess.run([os.path.join(BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, f)], capture_output=True, text=True).stdout
        for m in re.finditer(r"^[0-9a-f]+ <(\S+)>:\n(.*?)(?=^\n|^[0-9a-f]+ <[^>L][^>]*>:|\Z)", txt, re.S | re.M):
            pass
        syms = re.split(r"^[0-9a-f]+ <([^>]+)>:\n", txt, flags=re.M)
        # syms: [pre, name1, body1, name2, body2, ...]; labels (<L123>) are separate entries: glue them to their kernel
        cur, bodies = None, collections.OrderedDict()
        for i in range(1, len(syms), 2):
            n, b = syms[i], syms[i + 1]
            if n.startswith("L") and n[1:].isdigit() and cur:
                bodies[cur] += "<%s>:\n" % n + b
            else:
                cur = n
                bodies[cur] = b
        for n, b in bodies.items():
            if not pat.search(n):
                continue
            if len(sys.argv) > 2:
                open(sys.argv[2], "w").write(b)
            print(n)
            blocks = re.split(r"^<(L\d+)>:\n", b, flags=re.M)
            items = [("entry", blocks[0])] + [(blocks[i], blocks[i + 1]) for i in range(1, len(blocks), 2)]
            tot = collections.Counter()
            for lab, body in items:
                c = collections.Counter()
                for line in body.splitlines():
                    w = line.split()
                    if not w:
                        continue
                    op = w[0]
                    k = ("MFMA" if op.startswith("v_mfma") else "LDSr" if op.startswith("ds_read") or op.startswith("ds_bpermute") else "LDSw" if op.startswith("ds_") else
                         "VMEM" if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else "VALU" if op.startswith("v_") else
                         "WAIT" if op.startswith("s_waitcnt") else "BAR" if op.startswith("s_barrier") else "SALU" if op.startswith("s_") else "other")
                    c[k] += 1
                tot.update(c)
                if sum(c.values()) >= 40:
                    print("  %-8s %s" % (lab, "  ".join("%s %d" % kv for kv in sorted(c.items()))))
            print("  total    %s" % "  ".join("%s %d" % kv for kv in sorted(tot.items())))
            sys.exit(0)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
