#!/bin/bash
# One GPU call that re-checks a round: the gpu suite, the one-step rollout trace, the profile set (tools/profile_round.sh), the update
# trace and a plain bench run.  usage (through gpurun): bash tools/round_check.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/round_check; rm -rf $O; mkdir -p $O
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 6 $O/pytest.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 bash tools/rollout_trace.sh > $O/rollout_one_step_trace.txt 2>&1; tail -n 22 $O/rollout_one_step_trace.txt | cut -c1-120
timeout -k 10 1000 bash tools/profile_round.sh > $O/profile_round.log 2>&1; echo "profile rc=$?"; tail -n 4 $O/profile_round.log | cut -c1-300
timeout -k 10 300 bash tools/update_trace.sh > $O/update_trace.log 2>&1; tail -n 3 $O/update_trace.log | cut -c1-200
timeout -k 10 500 python3 bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/round_check/bench.json') if l.startswith('{')][-1])
r=d["roofline"]
print({k: d[k] for k in ("value","ms_per_step","rollout_env_steps_per_s","rollout_fraction_of_time","mean_episode_return")})
print({k: r[k] for k in ("frac","us_per_launch","profiler_child","us_per_launch_by_events_bound_to_the_dispatch","frac_standalone","us_per_launch_standalone")})
print(d["cpu_baseline"])
PY
