#!/usr/bin/env python3
"""mean per-dispatch counter values of the fa_* kernels from a rocprofv3 --pmc csv directory"""
import csv, glob, collections, sys
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "fa_" in k or "quantize" in k or "bwd" in k or "cast_rows" in k:
                acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in acc.items():
            print(k, {n: round(sum(v) / len(v) / 1e6, 5) for n, v in sorted(c.items())}, "n", len(next(iter(c.values()))))
