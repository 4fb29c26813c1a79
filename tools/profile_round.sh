#!/bin/bash
# Round profile set (run on the GPU box; outputs under gpurun_out/prof/, copy the summaries into profiles/ as rNN_*):
#  1. kernel stats of the bench (rocprofv3 --kernel-trace --stats)
#  2. the fused post-step kernel STANDALONE (tools/bench_kernels.py --post --plain: the product library, ONE launch variant per process):
#     kernel stats, FETCH_SIZE / WRITE_SIZE in separate --pmc passes, SQ counters - on boxes_64clips (L2-resident inputs, BASELINE
#     configs[2]) and on iter0_1024clips (configs[3] stand-in: 83 MB of clip rows + a 9 MB heightfield).  pmc_summary / pmc_traffic
#     refuse a pass in which the launches are not all of one shape (SQ_WAVES == 1024 workgroups x 6 waves).
#  3. the rollout loop's launches issued eagerly (tools/rollout_post_stats.sh): the in-rollout duration of every kernel
#  4. SQ counters of the simulator kernel (tools/bench_sim.py --plain)
# Counters always in their own passes, never combined with a trace domain other than --kernel-trace.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py --steps 3 --warmup 1 > $O/bench_n1.json.log 2>&1 &&
find $O/bench -name "*kernel_stats.csv" -exec cp {} $O/bench_n1_kernel_stats.csv \; || exit 1
rm -rf $O/bench
for W in boxes_64clips iter0_1024clips; do
  T=post_step_$W
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/post -- python3 tools/bench_kernels.py --post --plain --workload=$W > $O/$T.log 2>&1 &&
  find $O/post -name "*kernel_stats.csv" -exec cp {} $O/${T}_kernel_stats.csv \; &&
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcF -- python3 tools/bench_kernels.py --post --plain --workload=$W > /dev/null 2>&1 &&
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcW -- python3 tools/bench_kernels.py --post --plain --workload=$W > /dev/null 2>&1 &&
  python3 tools/pmc_traffic.py $O/pmcF $O/pmcW 4096 $W > $O/${T}_pmc_traffic.json &&
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $O/pmcA -- python3 tools/bench_kernels.py --post --plain --workload=$W > /dev/null 2>&1 &&
  python3 tools/pmc_summary.py $O/pmcA track_post --expect-waves=6144 > $O/${T}_sq_counters.json || exit 1
  rm -rf $O/post $O/pmcF $O/pmcW $O/pmcA
done
bash tools/rollout_post_stats.sh 96 > $O/rollout_eager.log 2>&1 && cp gpurun_out/rollpost_eager/rollout_kernel_stats.csv $O/rollout_kernel_stats.csv &&
cp gpurun_out/rollpost_eager/post_step_in_rollout.json $O/post_step_in_rollout.json || exit 1
bash tools/pmc_sim.sh > $O/sim_step_sq_counters.log 2>&1 && cp gpurun_out/sim_step_sq_counters.json $O/ || exit 1
tail -n 1 $O/bench_n1.json.log | cut -c1-600
ls -la $O
