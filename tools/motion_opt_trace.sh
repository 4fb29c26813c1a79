#!/bin/bash
# kernel statistics of the motion optimiser probe: 300 iterations, replayed graph only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PROBE_ONLY=graph
O=gpurun_out/motrace
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r -- python3 tools/motion_opt_probe.py 300 > $O/log.txt 2>&1
rc=$?
find $O/r -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/r
head -25 $O/kernel_stats.csv | cut -c1-200
exit $rc
