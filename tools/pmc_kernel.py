#!/usr/bin/env python
"""Average PMC counter values per kernel from rocprofv3 --pmc csv output directories.
usage: python tools/pmc_kernel.py <kernel-name-substring> <dir> [<dir> ...]"""
import csv, glob, sys, collections
flt = sys.argv[1]
for d in sys.argv[2:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if flt in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0][:60]][(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for k, v in acc.items():
        per = collections.defaultdict(list)
        for (c, _), x in v.items():
            per[c].append(x)
        print(d.split("/")[-1], k, {c: round(sum(x) / len(x)) for c, x in sorted(per.items())}, "launches", max(len(x) for x in per.values()))
