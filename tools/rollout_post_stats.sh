#!/bin/bash
# in-situ duration of the fused post-step launch: rocprofv3 kernel trace of the rollout loop, the track_post_kernel launch that follows
# each simulator step (the full launch; the masked restart launches are listed separately)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/rollpost
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/r -- python3 tools/rollout_only.py ${1:-96} > $O/log.txt 2>&1 &&
find $O/r -name "*kernel_trace.csv" -exec cp {} $O/trace.csv \;
rm -rf $O/r
python3 - <<'PY'
import csv, json, statistics as st
rows=list(csv.DictReader(open('gpurun_out/rollpost/trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
full,masked,sim=[],[],[]
prev_sim=False
for r in rows:
    n=r['Kernel_Name']; d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    if n.startswith('sim_step_bpl'): sim.append(d); prev_sim=True; continue
    if n.startswith('track_post_kernel'):
        (full if prev_sim else masked).append(d); prev_sim=False
    elif not n.startswith('void at::native::(anonymous namespace)::distribution'): pass
half=len(full)//2
out={"full_launches":len(full),"full_us_mean_second_half":round(st.mean(full[half:]),2),"full_us_median":round(st.median(full[half:]),2),
     "full_us_min":round(min(full),2),"masked_restart_launch_us_mean":round(st.mean(masked[len(masked)//2:]),2) if masked else None,
     "sim_step_us_mean":round(st.mean(sim[len(sim)//2:]),2)}
print(json.dumps(out))
PY
rm -f $O/trace.csv
