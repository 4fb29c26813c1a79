#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace csv: per (kernel, grid) count/min/median/mean duration in ns."""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    files = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)
    by = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("(")[0][-60:]
                by[(name, int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("%-62s %9s %5s %7s %8s %8s %8s" % ("kernel", "grid", "wg", "calls", "min_ns", "med_ns", "mean_ns"))
    for k, v in sorted(by.items()):
        v.sort()
        print("%-62s %9d %5d %7d %8d %8d %8d" % (k[0], k[1], k[2], len(v), v[0], v[len(v) // 2], sum(v) / len(v)))


if __name__ == "__main__":
    main()
