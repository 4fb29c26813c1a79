#!/usr/bin/env python3
"""HBM traffic of track_post_kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kilobytes per dispatch), with the
gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE under-reports wide coalesced reads by 2x).
usage: pmc_traffic.py <dir FETCH pass> <dir WRITE pass> <envs> [workload]"""
import collections, csv, glob, json, sys


def per_dispatch(d, counter, pat="track_post"):
    v = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"] and r["Counter_Name"] == counter:
                v.append(float(r["Counter_Value"]))
    v.sort()
    return {"launches": len(v), "mean_KB": sum(v) / len(v), "median_KB": v[len(v) // 2], "min_KB": v[0], "max_KB": v[-1]}


def main():
    fd, wd, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    f, w = per_dispatch(fd, "FETCH_SIZE"), per_dispatch(wd, "WRITE_SIZE")
    # one launch variant only: the kernel's stores are the same bytes every launch (round 3's files blended two variants: min 16.6 MB
    # against a median of 27.4 MB)
    if w["min_KB"] < 0.9 * w["median_KB"] or w["max_KB"] > 1.1 * w["median_KB"]:
        sys.exit("pmc_traffic: WRITE_SIZE spreads from %.0f to %.0f KB around a median of %.0f KB: more than one launch variant under the "
                 "profiler" % (w["min_KB"], w["max_KB"], w["median_KB"]))
    out = {"FETCH_SIZE": f, "WRITE_SIZE": w, "envs": n, "workload": sys.argv[4] if len(sys.argv) > 4 else "boxes_64clips",
           "note": "rocprofv3 --kernel-trace --pmc <counter> (separate passes) -- python3 tools/bench_kernels.py --post --plain --workload=<workload>; "
                   "track_post_kernel; bytes = KB*1024; gfx950 correction (MI355X_MICROARCH.md HBM section): corrected traffic = "
                   "(2*FETCH_SIZE + WRITE_SIZE)*1024",
           "traffic_bytes_corrected": (2 * f["mean_KB"] + w["mean_KB"]) * 1024,
           "traffic_bytes_raw": (f["mean_KB"] + w["mean_KB"]) * 1024,
           "algorithmic_bytes": n * (3544 + 7 * 760 + 456 + 3484 + 780 + 8)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
