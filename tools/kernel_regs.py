#!/usr/bin/env python3
"""VGPR / SGPR / LDS / scratch figures of the kernels in the built library whose name contains a substring (code-object metadata)."""
import re, subprocess, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_check as I
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesfmmm_amd", "libbfmmm_hip.so")
sub = sys.argv[1] if len(sys.argv) > 1 else "k_"
for co in I.extract_code_objects(lib, "/tmp/bfmmm_regs"):
    txt = subprocess.run([f"{I.LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name or sub not in name.group(1): continue
        g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "?"])[1]
        print(f"{name.group(1)[:90]:90s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>3s} sgpr {g('sgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
