#!/usr/bin/env python3
"""Per-kernel statistics of a rollout-loop kernel trace (rocprofv3 --kernel-trace --output-format csv of tools/rollout_only.py).

rocprofv3's own stats file has one row per kernel NAME, and track_post_kernel is launched in two shapes per step: the full launch
behind the simulator (all envs: observation + reward + termination) and the masked restart launch (finished envs only).  This keeps
them apart.  Used by tools/rollout_post_stats.sh (-> profiles/rNN_rollout_kernel_stats.csv) and by bench.py (roofline.us_per_launch).

usage: rollout_trace_stats.py <kernel_trace.csv> [<out stats csv>]   -> one JSON line (the full launches) on stdout"""
import csv
import json
import statistics as st
import sys

FULL = "track_post_kernel [full launch of the step, behind the simulator]"
MASKED = "track_post_kernel [masked restart launch]"


def stats(trace_csv, out_csv=None):
    rows = list(csv.DictReader(open(trace_csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by, sim = {}, []
    prev_sim = False
    for r in rows:
        n = r["Kernel_Name"]
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if n.startswith("sim_step_bpl"):
            prev_sim = True
            sim.append(d / 1e3)
        elif n.startswith("track_post_kernel"):
            n = FULL if prev_sim else MASKED
            prev_sim = False
        elif not (n.startswith("void at::native::(anonymous namespace)::distribution") or n.startswith("rng_step_kernel")):
            prev_sim = False                   # (a random-number launch may sit between the simulator and the post-step launch)
        by.setdefault(n, []).append(d)
    if out_csv:
        tot = sum(sum(v) for v in by.values())
        with open(out_csv, "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([n, len(v), sum(v), round(st.mean(v), 3), round(100.0 * sum(v) / tot, 2), min(v), max(v), round(st.pstdev(v), 3)])
    full = [d / 1e3 for d in by.get(FULL, [])]
    masked = [d / 1e3 for d in by.get(MASKED, [])]
    if not full:
        return None
    half = len(full) // 2
    return {"full_launches": len(full), "full_us_mean": round(st.mean(full), 3), "full_us_mean_second_half": round(st.mean(full[half:]), 3),
            "full_us_median": round(st.median(full), 3), "full_us_min": round(min(full), 3), "full_us_max": round(max(full), 3),
            "masked_restart_launch_us_mean": round(st.mean(masked), 3) if masked else None,
            "sim_step_us_mean": round(st.mean(sim), 3) if sim else None}


if __name__ == "__main__":
    print(json.dumps(stats(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)))
