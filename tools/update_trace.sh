#!/bin/bash
# ordered kernel list of ONE PPO minibatch (between two fused-loss launches), from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/updtrace
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/r -- python3 tools/update_only.py 1 > $O/log.txt 2>&1 &&
find $O/r -name "*kernel_trace.csv" -exec cp {} $O/trace.csv \;
rc=$?
rm -rf $O/r
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/updtrace/trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('ppo_sample')]
a,b=idx[-3],idx[-2]
t0=int(rows[a]['Start_Timestamp'])
small=0; smallt=0
with open('gpurun_out/updtrace/one_minibatch.txt','w') as f:
    for r in rows[a:b]:
        s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
        f.write("%8.1f %7.1f  %s\n"%((s-t0)/1e3,(e-s)/1e3,r['Kernel_Name'][:100]))
        if e-s<12000: small+=1; smallt+=e-s
    f.write("minibatch span us %.1f, kernels %d, kernels under 12 us: %d totalling %.1f us\n"%((int(rows[b]['Start_Timestamp'])-t0)/1e3,b-a,small,smallt/1e3))
PY
rm -f $O/trace.csv
exit $rc
