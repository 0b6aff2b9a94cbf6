"""Static check of the SHIPPED gfx950 code of k_sweep_chain (tools/isa_check.py): the chain wave prefetches C_a and the H rows two
steps ahead with inline-assembly `global_load_dwordx4` and hand-counted `s_waitcnt vmcnt(N)` (kernels_sweep.hip: sweep_ld16v,
swc_wait_c / swc_wait_h).  The compiler does not know that a prefetch destination is not valid until the counted wait, so it is
free to copy or read such a register in between -- this test disassembles the library that was built and fails if any
instruction touches a prefetch destination while its load can still be in flight, or if a counted wait is too weak for the loads
issued before it.  (Both happened while this test was written: a `v_mov_b64` of an H row ahead of its wait in a peeled last
step, and a last step that waited for a load it had not issued.)  CPU test: the code object is cross-compiled; no GPU needed."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_check as I  # noqa: E402

LIB = os.path.join(ROOT, "bayesfmmm_amd", "libbfmmm_hip.so")


def _ins(seq):
    """[(mnemonic, operands, branch target index or None)] -> the checker's instruction tuples (4-byte addresses)"""
    return [(4 * i, m, o, None if t is None else 4 * t) for i, (m, o, t) in enumerate(seq)]


def test_checker_known_answers():
    ld = lambda d, a="v2": ("global_load_dwordx4", f"v[{d}:{d + 3}], {a}, s[4:5]", None)
    # a copy of a destination before the wait
    v = I.check_function(_ins([ld(10), ("v_mov_b32_e32", "v20, v10", None), ("s_waitcnt", "vmcnt(0)", None), ("s_endpgm", "", None)]))
    assert len(v) == 1 and v[0][2] == [10]
    # after the wait: clean
    v = I.check_function(_ins([ld(10), ("s_waitcnt", "vmcnt(0)", None), ("v_mov_b32_e32", "v20, v10", None), ("s_endpgm", "", None)]))
    assert v == []
    # counted wait: vmcnt(1) retires the older load only; a store counts too
    seq = [ld(10), ld(14), ("s_waitcnt", "vmcnt(1)", None), ("v_add_f64", "v[30:31], v[10:11], v[12:13]", None),
           ("v_add_f64", "v[32:33], v[14:15], v[16:17]", None), ("s_endpgm", "", None)]
    v = I.check_function(_ins(seq))
    assert len(v) == 1 and v[0][2] == [14, 15, 16, 17]
    seq = [ld(10), ("global_store_dwordx2", "v2, v[40:41], s[4:5]", None), ("s_waitcnt", "vmcnt(1)", None),
           ("v_mov_b32_e32", "v20, v11", None), ("s_endpgm", "", None)]
    assert I.check_function(_ins(seq)) == []
    # a lgkmcnt-only wait does not retire vector-memory loads
    v = I.check_function(_ins([ld(10), ("s_waitcnt", "lgkmcnt(0)", None), ("v_mov_b32_e32", "v20, v10", None), ("s_endpgm", "", None)]))
    assert len(v) == 1
    # a loop that prefetches one trip ahead into alternating registers with a counted wait: clean;
    # the same loop with the wait one too weak: caught across the back edge
    def loop(n):
        return [ld(10), ld(14),                                    # 0, 1: prologue
                ("s_waitcnt", f"vmcnt({n})", None),               # 2: loop head
                ("v_mov_b32_e32", "v20, v10", None), ld(10),       # 3, 4: use A, refill A
                ("s_waitcnt", f"vmcnt({n})", None),
                ("v_mov_b32_e32", "v21, v14", None), ld(14),       # 6, 7: use B, refill B
                ("s_cbranch_scc1", "65530", 2),                    # 8: back edge
                ("s_waitcnt", "vmcnt(0)", None), ("s_endpgm", "", None)]
    assert I.check_function(_ins(loop(1))) == []
    assert len(I.check_function(_ins(loop(2)))) >= 1


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built")
def test_sweep_chain_prefetch_registers_are_not_touched_before_their_wait(tmp_path):
    res = I.check_library(LIB, "k_sweep_chain", str(tmp_path), markers=True)
    names = sorted(res)
    assert len(names) == 6 and all(f"k_sweep_chainILi{bw}E" in nm for bw, nm in enumerate(names)), names      # band half-widths 0 .. 5
    for nm in names:
        n_ins, viol = res[nm]
        assert n_ins > 1000
        assert viol == [], (nm, viol[:5])
