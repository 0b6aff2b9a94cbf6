"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean over launches).  Diagnostic helper.
usage: python tools/sq_counters.py DIR [DIR ...]"""
import collections
import csv
import glob
import sys

WANT = ("curve_chi", "curve_z", "sweep_chain", "k_factor", "pair_gram", "pg_reduce", "k_cov")
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("<")[0].split("(")[0][-24:]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            if any(x in k for x in WANT):
                print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
