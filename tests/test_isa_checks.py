"""Build-time ISA checks on the gfx950 code objects of libwtpse_hip.so (CPU: llvm-objdump, no GPU).

The BatchNorm-backward epilogues decide `fmaf(y, scale, shift) > 0` per element and must decide it exactly as the forward pass did.
Round 3 met a build in which the compiler had fused two such decisions into ONE `v_pk_fma_f32 ... op_sel_hi:[1,0,0]` (scale / shift
broadcast out of a register pair by the op_sel modifiers) and ~8 of 2.6 M masks, always in lanes 48-63, came out wrong in ~10 % of
the launches ("The v_pk_fma_f32 finding").  The ISA of that build (re-generated in round 4 from commit 6b61c00) has every
LDS read of the coefficients behind `s_waitcnt lgkmcnt(0)`, every load of y behind an in-order `vmcnt`, and at least one instruction
between the packed FMA and its consumers — no missing wait that a source-level fix could add — while the plain (no op_sel) packed FMAs
of the loaders' prologues run bit-exactly in every launch of every test.  So the guard is structural: the decision is kept scalar in
the source (an opaque value per element), and THIS test fails the build if any kernel that carries the epilogue contains a packed
fp32 FMA with operand-select modifiers, whatever a future compiler or edit makes of the source."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "wt-pse-code_amd", "wtpse_hip", "libwtpse_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

# template-argument positions of EPI in the mangled kernel names; EPI == 2 carries the BatchNorm-backward epilogue
EPI2 = [re.compile(r"^_Z9conv_x3_kILi\d+ELi\d+ELi\d+ELi2ELi\d+ELi\d+EE"),      # conv_x3_k<KS, MT, TWL, EPI, NT, TERMS>
        re.compile(r"^_Z10conv_x3r_kILi\d+ELi\d+ELi\d+ELi\d+ELi2ELi\d+EE"),     # conv_x3r_k<WM, MT, NT, TWL, EPI, TERMS>
        re.compile(r"^_Z10conv_fwd_kILi\d+ELi\d+ELi\d+ELb[01]ELi2EE")]          # conv_fwd_k<KS, MODE, TWL, DB, EPI>
ALSO = [re.compile(r"^_Z\d+maxpool2_bwd_bnb")]                                   # the max-pool backward with the same decision


@pytest.fixture(scope="module")
def disassembly():
    if not os.path.isfile(OBJDUMP) or not os.path.isfile(LIB):
        pytest.skip("needs llvm-objdump and the built library")
    tmp = tempfile.mkdtemp(prefix="wtpse_isa_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(LIB, so)
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


def test_bnb_epilogues_have_no_opsel_packed_fma(disassembly):
    checked = 0
    for name, body in disassembly.items():
        if not any(p.match(name) for p in EPI2 + ALSO):
            continue
        checked += 1
        bad = [l for l in body if "v_pk_fma_f32" in l and "op_sel" in l]
        assert not bad, "%s: packed fp32 FMA with operand-select modifiers in a kernel with the BatchNorm-backward epilogue:\n%s" % (
            name, "\n".join(bad[:4]))
    assert checked >= 12, "expected the EPI-2 instantiations of conv_x3_k, conv_x3r_k and conv_fwd_k in the library, found %d" % checked


def _broadcast_vgpr_operands(line):
    """Number of source operands of a v_pk_fma_f32 that are BROADCAST out of a VGPR pair: op_sel[i] == op_sel_hi[i] (both halves of
    the result read the same half of source i) and source i is a vector register pair."""
    m = re.match(r"v_pk_fma_f32\s+\S+,\s*(\S+),\s*(\S+),\s*(\S+?)(?:\s|$)", line)
    if not m:
        return 0
    srcs = [g.rstrip(",") for g in m.groups()]
    ms, mh = re.search(r"op_sel:\[([01]),([01]),([01])\]", line), re.search(r"op_sel_hi:\[([01]),([01]),([01])\]", line)
    sel = [int(x) for x in ms.groups()] if ms else [0, 0, 0]
    hi = [int(x) for x in mh.groups()] if mh else [1, 1, 1]
    return sum(1 for i in range(3) if sel[i] == hi[i] and re.match(r"^[va]\[", srcs[i]))


def test_no_kernel_has_a_two_broadcast_packed_fma(disassembly):
    """Round 5, VERDICT r04 weak 1: what distinguishes the instruction behind round 3's mask corruption from the 256 packed FMAs with
    an operand-select broadcast that ship (tools/isa_pkfma_census.py) is the NUMBER of operands broadcast out of VGPR pairs — two
    (`op_sel_hi:[1,0,0]` / `op_sel:[0,1,1]`: scale AND shift) against one everywhere in the shipped library (`op_sel_hi:[1,1,0]`: the
    addend of a prologue FMA).  A one-broadcast build of the failing epilogue is clean over 300 launches where the two-broadcast
    build fails in 16-36 (tools/probe/pk_variants.py v4 vs v0 / v2, profiles/r05_pk_fma_variants.txt).  So the guard is on the form:
    NO kernel of the library may contain a v_pk_fma_f32 with two or more VGPR-pair broadcasts, whatever the compiler makes of the
    source tomorrow."""
    n_pk = 0
    for name, body in disassembly.items():
        for l in body:
            if l.startswith("v_pk_fma_f32"):
                n_pk += 1
                assert _broadcast_vgpr_operands(l) < 2, "%s: packed fp32 FMA with %d VGPR-pair broadcasts: %s" % (name, _broadcast_vgpr_operands(l), l)
    assert n_pk > 500, "expected the library's packed FMAs in the disassembly, found %d" % n_pk


def test_x3_main_loops_are_spill_free(disassembly):
    """The MFMA kernels must not touch scratch memory (a register spill inside the MFMA stream costs more than any tuning gains)."""
    seen = 0
    for name, body in disassembly.items():
        if not re.match(r"^_Z(9conv_x3_k|10conv_x3r_k|9wgrad_r_k)I", name):
            continue
        if re.match(r"^_Z9wgrad_r_kILi2ELi2ELb[01]ELb[01]ELb1E", name):
            continue      # wtpse_conv_wgrad_r_bn's (2,2) block: known to spill, measured +10-22 % and not used by the step
        seen += 1
        assert not [l for l in body if "scratch_" in l], name
    assert seen > 10
