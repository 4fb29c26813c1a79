#!/bin/bash
# kernel mix of one rollout env step (graph replay): rocprofv3 kernel stats of tools/rollout_only.py, per-step counts and times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/rollmix
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r -- python3 tools/rollout_only.py 96 > $O/log.txt 2>&1 &&
find $O/r -name "*kernel_stats.csv" -exec cp {} $O/rollout_kernel_stats.csv \;
rc=$?
rm -rf $O/r
grep "rollout steps" $O/log.txt
exit $rc
