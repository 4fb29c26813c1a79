#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection csvs: mean per-dispatch value of every counter for kernels matching a pattern.
usage: pmc_summary.py <dir> <kernel-name substring> [--expect-waves=N]
--expect-waves: every dispatch must have SQ_WAVES == N (e.g. 1024 workgroups x 6 waves = 6144 for the full post-step launch at 4096
envs): a mean over launches of different shapes is not the figure of any of them (round 3's files blended two variants)."""
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
    expect = [int(a.split("=")[1]) for a in sys.argv[3:] if a.startswith("--expect-waves=")]
    if expect:
        w = by.get("SQ_WAVES", [])
        bad = [x for x in w if int(round(x)) != expect[0]]
        if not w or bad:
            sys.exit("pmc_summary: SQ_WAVES != %d in %d of %d dispatches (values seen: %s): more than one launch variant under the profiler"
                     % (expect[0], len(bad), len(w), sorted(set(int(round(x)) for x in w))[:8]))
    print(json.dumps({k: {"n": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in sorted(by.items())}, indent=1))


if __name__ == "__main__":
    main()
