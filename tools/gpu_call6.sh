#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c6; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/pytest.log
tail -n 15 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/probe/find_step_copies.py > $O/step_copies.txt 2>&1; grep "|" $O/step_copies.txt | sort | uniq -c | sort -rn | head -40
timeout -k 10 300 bash tools/rollout_post_stats.sh 96 --timed > $O/rollout_eager.log 2>&1; tail -n 3 $O/rollout_eager.log; grep "post-step launches timed" gpurun_out/rollpost_eager/log.txt
timeout -k 10 300 bash tools/rollout_trace.sh > $O/rollout_one_step_trace.txt 2>&1; tail -n 32 $O/rollout_one_step_trace.txt | cut -c1-140
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err; python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/c6/bench.json') if l.startswith('{')][-1])
r=d["roofline"]
print({k: d[k] for k in ("value","ms_per_step","rollout_env_steps_per_s","rollout_fraction_of_time","mean_episode_return")})
print({k: r[k] for k in ("frac","us_per_launch","us_per_launch_min","us_per_launch_max","frac_standalone","us_per_launch_standalone","launches_event_timed")})
for k in d["kernels"]: print(k["kernel"][:60], k.get("us_per_launch"))
PY
