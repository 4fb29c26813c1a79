#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection csvs: mean per-dispatch value of every counter for kernels matching a pattern."""
import collections
import csv
import glob
import json
import sys


def main():
    d, pat = sys.argv[1], sys.argv[2]
    by = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(json.dumps({k: {"n": len(v), "mean": sum(v) / len(v)} for k, v in sorted(by.items())}, indent=1))


if __name__ == "__main__":
    main()
