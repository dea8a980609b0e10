"""ISA lint of the shipped code objects (CPU test: disassembly only).

gfx950 erratum found in round 5 (profiles/r05_notes.md section 1; reproducers tools/debug/pkmul_repro.hip, pkmul_sweep.hip):
a packed-fp32 VALU instruction whose LOW result takes the HIGH dword of src1 while src0 contributes its low dword -
`v_pk_{mul,add,fma}_f32 ... op_sel:[0,1...]` with src0 != src1 - computes that low result with src1.hi read as 0 in lanes
48-63 whenever another wave of the same SIMD is executing one of the K-doubled 16x16 MFMAs of gfx950 (v_mfma_f32_16x16x32_bf16 / _f16, v_mfma_i32_16x16x64_i8; not 32x32x16, 16x16x16 or the fp32 16x16x4).  The
compiler emits the form by itself (SLP-vectorised scalar products); it was the non-reproducible gradient of round 4.  No code
object of libarco_hip.so may contain it."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "arco_amd", "lib", "libarco_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

_PK = re.compile(r"\b(v_pk_(?:mul|add|fma)_f32)\s+(\S+),\s*(\S+),\s*(\S+?)(?:,\s*(\S+))?\s+(.*)$")


def affected(line):
    """True for a packed-fp32 instruction of the erratum's form (see the module docstring)."""
    m = _PK.search(line.split("//")[0])
    if not m:
        return False
    mods = m.group(6)
    sel = re.search(r"op_sel:\[([01]),([01])", mods)
    if not sel or (sel.group(1), sel.group(2)) != ("0", "1"):
        return False
    return m.group(3) != m.group(4)          # the same register pair on both sources is not affected (measured)


def test_rule_on_examples():
    assert affected("v_pk_mul_f32 v[16:17], v[20:21], v[18:19] op_sel:[0,1] op_sel_hi:[1,0]// 0001AE90: D3B15010 08022514")
    assert affected("v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0]")
    assert affected("v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1] neg_lo:[0,1]")
    assert not affected("v_pk_add_f32 v[2:3], v[2:3], v[2:3] op_sel:[0,1] op_sel_hi:[1,0]")       # same pair
    assert not affected("v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,1]")
    assert not affected("v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[0,1]")
    assert not affected("v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1]")
    assert not affected("v_pk_mul_f32 v[38:39], v[22:23], v[34:35]")


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm image not present")
def test_no_code_object_holds_the_cross_select_packed_fp32_form(tmp_path):
    assert os.path.exists(LIB), "build the library first (__graft_entry__.build())"
    work = tmp_path / "lib"
    work.mkdir()
    shutil.copy(LIB, work / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=work, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = sorted(glob.glob(str(work / "*gfx950*")))
    assert len(objs) >= 7, objs           # one code object per translation unit with kernels
    hits, n_pk, fn = [], 0, "?"
    for o in objs:
        dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", o], check=True, capture_output=True, text=True).stdout
        for line in dis.splitlines():
            if line.endswith(">:"):
                fn = line.split("<")[-1][:-2]
            elif "v_pk_" in line and "_f32" in line:
                n_pk += 1
                if affected(line):
                    hits.append(f"{fn}: {line.strip()[:110]}")
    assert n_pk > 1000          # the scan saw the library's packed-fp32 code at all
    assert not hits, "gfx950 cross-select packed-fp32 erratum form found:\n" + "\n".join(hits)
