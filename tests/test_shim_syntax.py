"""The .Call shim (shim/bfmmm_rcall.cpp) is not built here (the image has no R), so it is at least type-checked on every run:
g++ -fsyntax-only against declaration-only stand-ins of the R API it uses (tools/rstub/, signatures from "Writing R Extensions"),
and the table it registers is compared with the reference's CallEntries[] (src/RcppExports.cpp:794-835: 33 symbols; names and
arities are listed here so that the test needs nothing outside the repository)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# name -> number of SEXP arguments, as registered by the reference package
CALL_ENTRIES = {
    "FMeanCI": 13, "HDFMeanCI": 12, "MVMeanCI": 6, "FCovCI": 15, "HDFCovCI": 14, "MVCovCI": 8, "SigmaCI": 4, "ZCI": 5,
    "FDIC": 10, "FAIC": 10, "FBIC": 10, "FLLik": 9, "MVAIC": 6, "MVBIC": 6, "MVDIC": 6, "MVLLik": 5,
    "ConditionalPredictiveOrdinates": 11, "FSamplePaths": 11,
    "BFMMM_Nu_Z_multiple_try": 28, "BFMMM_Theta_est": 32, "BFMMM_warm_start": 38,
    "ReadFieldCube": 1, "ReadFieldMat": 1, "ReadFieldVec": 1, "ReadCube": 1, "ReadMat": 1, "ReadVec": 1,
    "BHDFMMM_Nu_Z_multiple_try": 28, "BHDFMMM_Theta_est": 32, "BHDFMMM_warm_start": 38,
    "BMVMMM_Nu_Z_multiple_try": 23, "BMVMMM_Theta_est": 27, "BMVMMM_warm_start": 33,
}


def test_shim_type_checks_against_the_r_api_declarations():
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "tools", "rstub"),
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "shim", "bfmmm_rcall.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_shim_registers_the_reference_call_entries():
    src = open(os.path.join(ROOT, "shim", "bfmmm_rcall.cpp")).read()
    reg = dict((m.group(1), int(m.group(2))) for m in re.finditer(r'\{"_BayesFMMM_(\w+)",\s*\(DL_FUNC\)\s*&?_BayesFMMM_\w+,\s*(\d+)\}', src))
    assert reg == CALL_ENTRIES, (sorted(set(CALL_ENTRIES) ^ set(reg)), {k: (reg.get(k), v) for k, v in CALL_ENTRIES.items() if reg.get(k) != v})
    assert "R_init_BayesFMMM" in src and "R_useDynamicSymbols" in src
