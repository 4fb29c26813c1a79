#!/bin/bash
# What one PPO iteration launches OUTSIDE its 40 minibatches (build_train_data, buffer bookkeeping, copies): from a rocprofv3 kernel
# trace of tools/update_only.py 1, the launches between the last minibatch of the first update and the first minibatch of the
# second one, in order, and the per-name totals of the whole second update phase.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/itertrace
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/r -- python3 tools/update_only.py 1 > $O/log.txt 2>&1 &&
find $O/r -name "*kernel_trace.csv" -exec cp {} $O/trace.csv \;
rc=$?
rm -rf $O/r
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/itertrace/trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('ppo_sample')]
per_update = len(idx) // 2
a = idx[per_update - 1]                 # last minibatch of the first update
b = idx[per_update]                     # first minibatch of the second update
# the last minibatch ends with its optimiser step (one flat launch for both networks): the first sgd kernel after a
k = a
seen = 0
while seen < 1:
    k += 1
    seen += rows[k]['Kernel_Name'].startswith('sgd_momentum')
with open('gpurun_out/itertrace/between_updates.txt', 'w') as f:
    t0 = int(rows[k]['End_Timestamp'])
    tot = 0
    for r in rows[k + 1:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        tot += e - s
        f.write("%9.1f %8.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, r['Kernel_Name'][:110]))
    f.write("span us %.1f, launches %d, busy us %.1f\n" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, b - k - 1, tot / 1e3))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rows[b:]:
        n = r['Kernel_Name'][:90]
        agg[n][0] += 1
        agg[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    f.write("\n# second update phase (%d minibatches), totals by kernel name: launches, total us\n" % per_update)
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write("%6d %10.1f  %s\n" % (c, t / 1e3, n))
    f.write("phase span us %.1f\n" % ((int(rows[-1]['End_Timestamp']) - int(rows[b]['Start_Timestamp'])) / 1e3))
PY
rm -f $O/trace.csv
exit $rc
