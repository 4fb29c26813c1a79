#!/bin/bash
# Round profile set (run on the GPU box): kernel stats of the bench; kernel stats + PMC traffic (+ SQ counters) of the fused
# post-step kernel on the L2-resident workload (boxes_64clips, BASELINE configs[2]) and on the iter-0 stand-in (iter0_1024clips,
# configs[3]: 83 MB of clip rows + a 9 MB heightfield, nothing fits in L2).  Counters in their own passes (FETCH_SIZE and WRITE_SIZE
# do not fit one pass), never combined with a trace domain other than --kernel-trace.  Outputs under gpurun_out/prof/ ; copy the
# summaries into profiles/ as rNN_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py --steps 3 --warmup 1 > $O/bench_n1.json.log 2>&1 &&
find $O/bench -name "*kernel_stats.csv" -exec cp {} $O/bench_n1_kernel_stats.csv \; || exit 1
for W in boxes_64clips iter0_1024clips; do
  T=post_step_$W
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/post -- python3 tools/bench_kernels.py --post --plain --workload=$W > $O/$T.log 2>&1 &&
  find $O/post -name "*kernel_stats.csv" -exec cp {} $O/${T}_kernel_stats.csv \; &&
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcF -- python3 tools/bench_kernels.py --post --plain --workload=$W > /dev/null 2>&1 &&
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcW -- python3 tools/bench_kernels.py --post --plain --workload=$W > /dev/null 2>&1 &&
  python3 tools/pmc_traffic.py $O/pmcF $O/pmcW 4096 $W > $O/${T}_pmc_traffic.json &&
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $O/pmcA -- python3 tools/bench_kernels.py --post --plain --workload=$W > /dev/null 2>&1 &&
  python3 tools/pmc_summary.py $O/pmcA track_post > $O/${T}_sq_counters.json || exit 1
  rm -rf $O/post $O/pmcF $O/pmcW $O/pmcA
done
rm -rf $O/bench
tail -n 1 $O/bench_n1.json.log | cut -c1-400
ls -la $O
