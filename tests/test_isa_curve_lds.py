"""Static check of the SHIPPED gfx950 code of the per-curve kernels at the BASELINE config-2 shape (K = 3, M = 6, cubic splines),
against the three properties round 4 measured their speed to hang on (last part):

  * the row dot products and the band / theta reads go through single `ds_read_b64` -- the compiler pairs adjacent doubles into
    `ds_read2_b64`, which the LDS serves at half the rate (lds_dot.hpp: dot_lds, lds_ld); a rebuild that loses the hand-scheduled
    block or the opaque addresses shows up here as a `ds_read2_b64` count back in the hundreds;
  * the hand-scheduled dot keeps its counted waits (`s_waitcnt lgkmcnt(8)`: two chunks of eight reads in flight);
  * the exact instance of k_curve_chi stays within 128 VGPRs and the lean k_curve_z within 128: the fourth workgroup per CU.

CPU test: the code object is cross-compiled; no GPU needed."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_check as I  # noqa: E402

LIB = os.path.join(ROOT, "bayesfmmm_amd", "libbfmmm_hip.so")
CHI = "_ZN5bfmmm11k_curve_chiILi3ELi32ELb0ELb1ELi3ELi6EEEvNS_3CtxEi"      # <BW 3, 32 lanes, no covariates, SMALL, K 3, M 6>
LEANZ = "_ZN5bfmmm9k_curve_zILi3ELi32ELb0ELi3ELb1ELb1EEEvNS_3CtxEi"       # <BW 3, 32 lanes, no covariates, K 3, lean, exact K>


def _kernel(tmp, sym):
    """(disassembly text, metadata block) of one kernel of the built library"""
    for co in I.extract_code_objects(LIB, str(tmp)):
        txt = subprocess.run([f"{I.LLVM}/llvm-objdump", "-d", "--disassemble-symbols=" + sym, co], capture_output=True, text=True).stdout
        if sym in txt and len(txt) > 2000:
            notes = subprocess.run([f"{I.LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            blk = [b for b in notes.split("- .agpr_count:")[1:] if re.search(r"\.name:\s+" + re.escape(sym) + r"\s", b)]
            assert len(blk) == 1, sym
            return txt, blk[0]
    raise AssertionError("kernel not found in the library: " + sym)


def _count(txt, mnemonic):
    return len(re.findall(r"^\s+" + mnemonic + r"\b", txt, re.M))


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built")
@pytest.mark.parametrize("sym,max_read2,min_read1,min_waits", [(CHI, 40, 200, 10), (LEANZ, 40, 80, 3)])
def test_per_curve_kernels_read_lds_with_single_b64_loads(tmp_path, sym, max_read2, min_read1, min_waits):
    txt, meta = _kernel(tmp_path, sym)
    n2, n1 = _count(txt, "ds_read2_b64"), _count(txt, "ds_read_b64")
    assert n2 <= max_read2, (sym, "ds_read2_b64", n2)      # (what is left: the scalar job's tables and a few staging reads)
    assert n1 >= min_read1, (sym, "ds_read_b64", n1)
    # the hand-scheduled dots: 7 counted waits per 32-entry block (the chi forms), 3 per 16-entry half row (the Z forms, two lanes each)
    assert txt.count("s_waitcnt lgkmcnt(8)") >= min_waits, sym
    vg = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
    spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1))
    assert vg <= 128 and spill == 0, (sym, vg, spill)      # four workgroups (sixteen waves) per CU
