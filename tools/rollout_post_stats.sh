#!/bin/bash
# Kernel stats of the ROLLOUT loop (no PPO update), per kernel, from rocprofv3 --kernel-trace --stats:
#   eager launches (default): rocprofv3's begin/end stamps are the kernels' own - this is the in-rollout duration of every kernel, the
#                             post-step launch included, in the cache state the rollout leaves it (-> profiles/rNN_rollout_kernel_stats.csv)
#   --graph               : the product's hipGraph replays.  rocprofv3 adds ~3 us to EVERY kernel node of a replayed graph (a 0.6 us fill
#                             reads 4.2 us: profiles/r04_rocprof_graph_node_inflation.txt), so these durations are upper bounds only
# usage (GPU box): bash tools/rollout_post_stats.sh [--graph] [steps] [extra args of tools/rollout_only.py, e.g. --diag=0x100000]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MODE=--eager; TAG=eager
if [ "$1" == "--graph" ]; then MODE=""; TAG=graph; shift; fi
STEPS=${1:-96}; shift
O=gpurun_out/rollpost_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r -- python3 tools/rollout_only.py $STEPS $MODE "$@" > $O/log.txt 2>&1 &&
find $O/r -name "*kernel_trace.csv" -exec cp {} $O/trace.csv \;
rm -rf $O/r
python3 - $O <<'PY'
import csv, json, statistics as st, sys
O = sys.argv[1]
rows = list(csv.DictReader(open(O + '/trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
full, masked, sim = [], [], []
prev_sim = False
for r in rows:
    n = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if n.startswith('sim_step_bpl'):
        sim.append(d); prev_sim = True; continue
    if n.startswith('track_post_kernel'):
        (full if prev_sim else masked).append(d); prev_sim = False
    elif n.startswith('void at::native::(anonymous namespace)::distribution'):
        pass                       # the uniform pool between the simulator and the post-step launch
    else:
        prev_sim = False
half = len(full) // 2
# per-kernel stats in rocprofv3's own column layout, with the two launch shapes of track_post_kernel kept apart (rocprofv3's stats file
# has one row per kernel NAME: the full launch behind the simulator and the masked restart launch would be averaged together)
by = {}
prev_sim = False
for r in rows:
    n = r['Kernel_Name']; d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    if n.startswith('sim_step_bpl'):
        prev_sim = True
    elif n.startswith('track_post_kernel'):
        n = 'track_post_kernel [full launch of the step, behind the simulator]' if prev_sim else 'track_post_kernel [masked restart launch]'
        prev_sim = False
    elif not n.startswith('void at::native::(anonymous namespace)::distribution'):
        prev_sim = False
    by.setdefault(n, []).append(d)
tot = sum(sum(v) for v in by.values())
with open(O + '/rollout_kernel_stats.csv', 'w', newline='') as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([n, len(v), sum(v), round(st.mean(v), 3), round(100.0 * sum(v) / tot, 2), min(v), max(v), round(st.pstdev(v), 3)])
out = {"full_launches": len(full), "full_us_mean": round(st.mean(full), 2), "full_us_mean_second_half": round(st.mean(full[half:]), 2),
       "full_us_median": round(st.median(full[half:]), 2), "full_us_min": round(min(full), 2),
       "masked_restart_launch_us_mean": round(st.mean(masked[len(masked) // 2:]), 2) if masked else None,
       "sim_step_us_mean": round(st.mean(sim[len(sim) // 2:]), 2)}
print(json.dumps(out))
json.dump(out, open(O + '/post_step_in_rollout.json', 'w'))
PY
rm -f $O/trace.csv
tail -n 1 $O/log.txt
