#!/usr/bin/env python3
"""Static check of the gfx950 code of a kernel for ONE hazard the hardware does not interlock and the compiler cannot see when the
loads are issued by inline assembly (k_sweep_chain's hand-counted prefetch, kernels_sweep.hip `sweep_ld16v` / `swc_wait_*`):

    a vector register that is the destination of a vector-memory load still in flight (not yet covered by an `s_waitcnt vmcnt(N)`)
    must not be read, written or copied by any other instruction.

The checker disassembles the code object (llvm-objdump), builds the control-flow graph of the kernel and runs a forward data-flow
analysis whose state is the FIFO of outstanding vector-memory operations (gfx9 family: loads AND stores count in vmcnt and retire
in issue order; `s_waitcnt vmcnt(N)` leaves the N youngest in flight).  A block is analysed under every distinct FIFO that can reach
it (bounded), so the two-steps-ahead prefetch rotating through its register sets across loop iterations is followed exactly.

Used by tests/test_isa_sweep_chain.py (CPU test; no GPU needed: the code object is cross-compiled).
"""
import os
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin"
BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def extract_code_objects(lib_path, workdir):
    """the gfx950 code objects inside a HIP shared library / object file (one per translation unit)"""
    os.makedirs(workdir, exist_ok=True)
    fat = os.path.join(workdir, "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
    data = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(BUNDLE_MAGIC), data)]
    out = []
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else len(data)
        part = os.path.join(workdir, f"bundle{k}.bin")
        open(part, "wb").write(data[s:e])
        co = os.path.join(workdir, f"code{k}.co")
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", f"--targets={TARGET}",
                            f"--output={co}"], capture_output=True)
        if r.returncode == 0 and os.path.exists(co):
            out.append(co)
    return out


def disassemble_function(code_object, mangled_substr):
    """[(address, mnemonic, operand string, branch target or None)] of every function whose symbol contains `mangled_substr`"""
    txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", code_object], capture_output=True, text=True, check=True).stdout
    funcs, cur, base = {}, None, 0
    for line in txt.split("\n"):
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
        if m:
            name = m.group(2)
            cur = name if mangled_substr in name else None
            base = int(m.group(1), 16)
            if cur:
                funcs[cur] = []
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if not m:
            continue
        mnem, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        tgt = None
        if mnem.startswith("s_cbranch") or mnem == "s_branch":
            mt = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", line)
            tgt = base + int(mt.group(1), 16) if mt else (base if re.search(r"<[^>+]*>\s*$", line) else None)
        funcs[cur].append((addr, mnem, ops, tgt))
    return funcs


_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(opstr):
    s = set()
    for m in _VREG.finditer(opstr):
        if m.group(1) is not None:
            s.add(int(m.group(1)))
        else:
            s.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return s


def is_vmem(mnem):
    return mnem.startswith(("global_", "buffer_", "flat_", "scratch_", "tbuffer_"))


def vmem_dest(mnem, ops):
    """destination registers of a vector-memory instruction that returns data (first operand), else empty"""
    returns = ("_load" in mnem and "load_lds" not in mnem) or ("atomic" in mnem and re.search(r"\b(glc|sc0)\b", ops) is not None)
    if returns:
        return frozenset(vregs(ops.split(",")[0]))
    return frozenset()


def vmcnt_of(ops):
    """N of `s_waitcnt ... vmcnt(N)`; None when the instruction does not wait on vmcnt.  A raw immediate is decoded (gfx9 layout:
    vmcnt = bits 3:0 | bits 15:14 << 4)."""
    m = re.search(r"vmcnt\((\d+)\)", ops)
    if m:
        return int(m.group(1))
    if re.fullmatch(r"(0x[0-9a-fA-F]+|\d+)", ops.strip()):
        imm = int(ops.strip(), 0)
        v = (imm & 0xF) | (((imm >> 14) & 0x3) << 4)
        return None if v == 63 else v
    return None


def check_function(insns, max_states=256, max_fifo=64):
    """returns a list of violations [(address, text, offending registers, address of the load)]"""
    if not insns:
        return [("-", "function not found", set(), "-")]
    addr_index = {a: i for i, (a, _, _, _) in enumerate(insns)}
    leaders = {0}
    for i, (a, mnem, ops, tgt) in enumerate(insns):
        if tgt is not None:
            if tgt in addr_index:
                leaders.add(addr_index[tgt])
            if i + 1 < len(insns):
                leaders.add(i + 1)
        if mnem in ("s_endpgm", "s_setpc_b64", "s_swappc_b64") and i + 1 < len(insns):
            leaders.add(i + 1)
    leaders = sorted(leaders)
    block_of = {}
    blocks = []
    for bi, s in enumerate(leaders):
        e = leaders[bi + 1] if bi + 1 < len(leaders) else len(insns)
        blocks.append((s, e))
        block_of[s] = bi
    violations = {}
    seen = [set() for _ in blocks]
    work = [(0, ())]
    seen[0].add(())
    while work:
        bi, fifo = work.pop()
        s, e = blocks[bi]
        fifo = list(fifo)
        fall = True
        for i in range(s, e):
            a, mnem, ops, tgt = insns[i]
            if mnem == "s_waitcnt":
                n = vmcnt_of(ops)
                if n is not None and len(fifo) > n:
                    fifo = fifo[len(fifo) - n:] if n > 0 else []
                continue
            touched = vregs(ops)
            if touched:
                for (la, regs) in fifo:
                    bad = regs & touched
                    if bad:
                        violations.setdefault((a, la), (hex(a), f"{mnem} {ops}", sorted(bad), hex(la)))
            if is_vmem(mnem):
                fifo.append((a, vmem_dest(mnem, ops)))
                if len(fifo) > max_fifo:
                    fifo = fifo[-max_fifo:]
            if mnem == "s_endpgm":
                fall = False
        last = insns[e - 1]
        succ = []
        if last[3] is not None and last[3] in addr_index:
            succ.append(block_of[addr_index[last[3]]])
        if last[1] == "s_branch" or last[1] in ("s_endpgm", "s_setpc_b64"):
            fall = False
        if fall and e < len(insns):
            succ.append(block_of[e])
        key = tuple(fifo)
        for sb in succ:
            if key not in seen[sb]:
                if len(seen[sb]) >= max_states:
                    violations.setdefault(("states", sb), (hex(insns[blocks[sb][0]][0]), "too many distinct in-flight states reach this block "
                                                           "(analysis bound)", [], "-"))
                    continue
                seen[sb].add(key)
                work.append((sb, key))
    return list(violations.values())


def marker_ranges(insns, begin, end):
    """address ranges [a_begin, a_end] delimited by the marker instructions `s_nop <begin>` / `s_nop <end>`"""
    out, start = [], None
    for a, mnem, ops, _ in insns:
        if mnem == "s_nop" and ops.strip() == str(begin):
            start = a
        elif mnem == "s_nop" and ops.strip() == str(end) and start is not None:
            out.append((start, a))
            start = None
    return out


def filter_by_markers(insns, violations, load_markers=((13, 14), (11, 12)), touch_markers=((11, 12),)):
    """keep the violations whose LOAD lies in a `load_markers` range and whose touching instruction lies in a `touch_markers`
    range.  The data-flow analysis is path-insensitive: a kernel whose waves take wave-uniform but mutually exclusive branches
    (k_sweep_chain: the chain wave / the row threads) shows harmless "violations" between the two branches -- registers of one
    branch reused by the other under a disjoint EXEC mask.  The markers bracket the one place where the hazard is real: the
    inline-assembly prefetch of the chain wave (prologue loads: 13..14; the chain region: 11..12)."""
    lr = [r for b, e in load_markers for r in marker_ranges(insns, b, e)]
    tr = [r for b, e in touch_markers for r in marker_ranges(insns, b, e)]
    inside = lambda a, rs: any(lo <= a <= hi for lo, hi in rs)
    keep = []
    for v in violations:
        try:
            ta, la = int(v[0], 16), int(v[3], 16)
        except (ValueError, TypeError):
            keep.append(v)
            continue
        if inside(ta, tr) and inside(la, lr):
            keep.append(v)
    return keep, lr, tr


def check_library(lib_path, mangled_substr, workdir, markers=False):
    res = {}
    for co in extract_code_objects(lib_path, workdir):
        for name, insns in disassemble_function(co, mangled_substr).items():
            v = check_function(insns)
            if markers:
                v, lr, tr = filter_by_markers(insns, v)
                if not lr or not tr:
                    v = [("-", "marker instructions (s_nop 11/12/13/14) not found", [], "-")]
            res[name] = (len(insns), v)
    return res


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesfmmm_amd", "libbfmmm_hip.so")
    sub = sys.argv[2] if len(sys.argv) > 2 else "k_sweep_chain"
    r = check_library(lib, sub, "/tmp/bfmmm_isa_check", markers=(sub == "k_sweep_chain"))
    bad = 0
    for name, (n, v) in sorted(r.items()):
        print(f"{name}: {n} instructions, {len(v)} violation(s)")
        for x in v[:20]:
            print("   ", x)
        bad += len(v)
    sys.exit(1 if bad or not r else 0)
