#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c7; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 12 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
cd /tmp; timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rf -- python3 $GRAFT_REPO_ROOT/tools/probe/replay_fill.py > /dev/null 2>&1; grep -h "FillFunctor\|Functor_add" $(find /tmp/rf -name "*kernel_stats.csv") | cut -c1-60,140-200; cd $GRAFT_REPO_ROOT
timeout -k 10 300 bash tools/rollout_trace.sh > $O/rollout_one_step_trace.txt 2>&1; tail -n 28 $O/rollout_one_step_trace.txt | cut -c1-140
timeout -k 10 500 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -n 3 $O/bench.err; python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/c7/bench.json') if l.startswith('{')][-1])
r=d["roofline"]
print({k: d[k] for k in ("value","ms_per_step","rollout_env_steps_per_s","rollout_fraction_of_time","mean_episode_return")})
print({k: r[k] for k in ("frac","us_per_launch","us_per_launch_source","profiler_child","us_per_launch_by_events_bound_to_the_dispatch","frac_standalone","us_per_launch_standalone")})
PY
