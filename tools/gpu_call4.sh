#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c4; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/pytest.log
tail -n 12 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 bash tools/rollout_post_stats.sh 96 > $O/rollout_eager.log 2>&1; tail -n 2 $O/rollout_eager.log
timeout -k 10 300 bash tools/rollout_post_stats.sh --graph 96 > $O/rollout_graph.log 2>&1; tail -n 2 $O/rollout_graph.log
timeout -k 10 300 python3 tools/sim_variants.py run > $O/sim_variants.txt 2>&1; cat $O/sim_variants.txt | grep variant
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
