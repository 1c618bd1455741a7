#!/usr/bin/env python3
"""Build the four variants of the round-3 BatchNorm-backward epilogue that tools/probe/pk_variants.py times (BUILD CONTAINER ONLY: it
reads the sources of commit 6b61c00 from this repository's history and compiles them with hipcc into tools/probe/_build/, which is
git-ignored but travels to the GPU box).

    v0  the epilogue as it failed: no guard — the compiler fuses the two ReLU decisions of a register into one
        v_pk_fma_f32 ... op_sel_hi:[1,0,0]
    v1  the shipped guard: an opaque value keeps the decision a scalar v_fma_f32
    v2  v0 with every output store moved behind all decisions
    v3  an explicit packed FMA on real register pairs (opaque copies of scale / shift): v_pk_fma_f32 without op_sel modifiers
    v4  (round 5) ONE operand broadcast: the scale a real register pair (opaque copy), the shift left to the compiler's operand-select
        broadcast — the form the shipped wgrad_r_k / head_bwd_k prologues contain (tools/isa_pkfma_census.py: 256 sites, none with two)
Prints, per variant, how many packed FMAs (and how many with operand-select modifiers) the EPI-2 kernel conv_x3_k<3,2,5,2,2> contains."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REV = "6b61c00"
FILES = ("conv_x3.hip", "common.h", "conv.hip", "bn.hip")

OLD = '''          float zz = __builtin_fmaf(mk[r][nt], bsc, bsh);
          asm volatile("" : "+v"(zz));
          if (!(zz > 0.f)) v = 0.f;
          acc[mt][nt][r] = v;
        }
        buf_store(rs_o, pvo[nt], soff, v);
      }
    }'''
V0 = OLD.replace('          asm volatile("" : "+v"(zz));\n', '')
V2 = '''          float zz = __builtin_fmaf(mk[r][nt], bsc, bsh);
          if (!(zz > 0.f)) v = 0.f;
          acc[mt][nt][r] = v;
        }
        if (!BNB) buf_store(rs_o, pvo[nt], soff, v);
      }
    }
    if (BNB) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < NACC; ++r) {
        const int cbase = cout0 + mt * 32 + (r & 3) + 8 * (r >> 2);
        const bool second = a.out1 != nullptr && cbase >= a.Csplit;
        const __amdgpu_buffer_rsrc_t rs_o = second ? rs_o1 : rs_o0;
        const unsigned soff = (unsigned)(second ? min(cbase, a.Cout) - a.Csplit : min(cbase, a.Csplit)) * hw4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) buf_store(rs_o, pvo[nt], soff, acc[mt][nt][r]);
      }
    }'''
HEAD3 = '''#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float v = fmaxf(acc[mt][nt][r], relu_lo);
        if (MASK && !(mk[r][nt] > 0.f)) v = 0.f;
        if (BNB) {     // the ReLU decision of the forward pass: fmaf(y, scale, shift) > 0 (channels outside [bn_c0, bn_c1): 0, 1)'''
NEW4 = '''      typedef float f32x2v __attribute__((ext_vector_type(2)));
      f32x2v zpk = {0.f, 0.f};
      if (BNB && NT == 2) {
        f32x2v mm2 = {mk[r][0], mk[r][1]}, sc2 = {bsc, bsc}, sh2 = {bsh, bsh};
        asm volatile("" : "+v"(sc2));
        zpk = __builtin_elementwise_fma(mm2, sc2, sh2);
      }
''' + HEAD3
NEW3 = '''      typedef float f32x2v __attribute__((ext_vector_type(2)));
      f32x2v zpk = {0.f, 0.f};
      if (BNB && NT == 2) {
        f32x2v mm2 = {mk[r][0], mk[r][1]}, sc2 = {bsc, bsc}, sh2 = {bsh, bsh};
        asm volatile("" : "+v"(sc2), "+v"(sh2));
        zpk = __builtin_elementwise_fma(mm2, sc2, sh2);
      }
''' + HEAD3


def main():
    out = os.path.join(HERE, "_build")
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        for f in FILES:
            src = subprocess.check_output(["git", "-C", ROOT, "show", "%s:wt-pse-code_amd/wtpse_hip/csrc/%s" % (REV, f)], text=True)
            open(os.path.join(tmp, f), "w").write(src)
        base = open(os.path.join(tmp, "conv_x3.hip")).read()
        assert OLD in base and HEAD3 in base
        v3 = base.replace(HEAD3, NEW3).replace('''          float zz = __builtin_fmaf(mk[r][nt], bsc, bsh);
          asm volatile("" : "+v"(zz));''', '''          float zz = NT == 2 ? zpk[nt] : __builtin_fmaf(mk[r][nt], bsc, bsh);''')
        v4 = base.replace(HEAD3, NEW4).replace('''          float zz = __builtin_fmaf(mk[r][nt], bsc, bsh);
          asm volatile("" : "+v"(zz));''', '''          float zz = NT == 2 ? zpk[nt] : __builtin_fmaf(mk[r][nt], bsc, bsh);''')
        variants = {0: base.replace(OLD, V0), 1: base, 2: base.replace(OLD, V2), 3: v3, 4: v4}
        for v, text in variants.items():
            p = os.path.join(tmp, "conv_x3_v%d.hip" % v)
            open(p, "w").write(text)
            common = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", tmp]
            subprocess.check_call(common + ["-shared", p, os.path.join(tmp, "conv.hip"), os.path.join(tmp, "bn.hip"), "-o",
                                            os.path.join(out, "libpk_v%d.so" % v)], stderr=subprocess.DEVNULL)
            asm = subprocess.check_output(common + ["-S", "--cuda-device-only", p, "-o", "-"], stderr=subprocess.DEVNULL, text=True)
            body, on = [], False
            for line in asm.splitlines():
                if line.startswith("_Z9conv_x3_kILi3ELi2ELi5ELi2ELi2EEv10ConvX3Args:"):
                    on = True
                if on:
                    body.append(line)
                    if "s_endpgm" in line:
                        break
            pk = [l for l in body if "v_pk_fma_f32" in l]
            import re
            nb = []
            for l in pk:
                ms, mh = re.search(r"op_sel:\[([01]),([01]),([01])\]", l), re.search(r"op_sel_hi:\[([01]),([01]),([01])\]", l)
                sel = [int(x) for x in ms.groups()] if ms else [0, 0, 0]
                hi = [int(x) for x in mh.groups()] if mh else [1, 1, 1]
                nb.append(sum(a == b for a, b in zip(sel, hi)))
            print("v%d: %d v_pk_fma_f32 in conv_x3_k<3,2,5,2,2>, %d of them with op_sel modifiers; broadcast operands per instruction: %s; e.g. %s"
                  % (v, len(pk), sum("op_sel" in l for l in pk), sorted(set(nb)), pk[0].strip() if pk else "-"))


if __name__ == "__main__":
    main()
