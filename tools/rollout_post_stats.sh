#!/bin/bash
# Kernel stats of the ROLLOUT loop (no PPO update), per kernel, from a rocprofv3 --kernel-trace:
#   eager launches (default): the step's launch sequence issued eagerly (tools/rollout_only.py --eager) - the in-rollout duration of every
#                             kernel, the post-step launch included, in the cache state the rollout leaves it (-> profiles/rNN_rollout_kernel_stats.csv)
#   --graph                 : the product's hipGraph replays.  rocprofv3 adds ~3 us to every SMALL kernel node of a replayed graph (a 0.6 us
#                             fill reads 4.2 us: profiles/r04_rocprof_graph_node_inflation.txt); large nodes read the same as eagerly
# usage (GPU box): bash tools/rollout_post_stats.sh [--graph] [steps] [extra args of tools/rollout_only.py, e.g. --timed]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MODE=--eager; TAG=eager
if [ "$1" == "--graph" ]; then MODE=""; TAG=graph; shift; fi
STEPS=${1:-96}; shift
O=gpurun_out/rollpost_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/r -- python3 tools/rollout_only.py $STEPS $MODE "$@" > $O/log.txt 2>&1 &&
find $O/r -name "*kernel_trace.csv" -exec cp {} $O/trace.csv \;
rm -rf $O/r
python3 tools/rollout_trace_stats.py $O/trace.csv $O/rollout_kernel_stats.csv | tee $O/post_step_in_rollout.json
rm -f $O/trace.csv
tail -n 2 $O/log.txt
