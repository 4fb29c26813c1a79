#!/bin/bash
# ordered kernel list of ONE rollout env step (between two simulator launches), from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/rolltrace
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/r -- python3 tools/rollout_only.py 24 > $O/log.txt 2>&1 &&
find $O/r -name "*kernel_trace.csv" -exec cp {} $O/trace.csv \;
rc=$?
rm -rf $O/r
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/rolltrace/trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('sim_step_bpl')]
a,b=idx[-3],idx[-2]
t0=int(rows[a]['Start_Timestamp'])
with open('gpurun_out/rolltrace/one_step.txt','w') as f:
    for r in rows[a:b]:
        s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
        f.write("%8.1f %6.1f  %s\n"%((s-t0)/1e3,(e-s)/1e3,r['Kernel_Name'][:110]))
    f.write("step span us %.1f, kernels %d\n"%((int(rows[b]['Start_Timestamp'])-t0)/1e3,b-a))
PY
cat $O/one_step.txt; rm -f $O/trace.csv
exit $rc
