#!/bin/bash
# Round profile set (run on the GPU box): kernel stats of the bench, kernel stats + PMC traffic + SQ counters of the fused
# post-step kernel.  Outputs under gpurun_out/prof/ ; copy the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 bench.py --steps 3 --warmup 1 > $O/bench_n1.json.log 2>&1 &&
find $O/bench -name "*kernel_stats.csv" -exec cp {} $O/bench_n1_kernel_stats.csv \; &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/post -- python3 tools/bench_kernels.py --post --plain > $O/post.log 2>&1 &&
find $O/post -name "*kernel_stats.csv" -exec cp {} $O/post_step_kernel_stats.csv \; &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcF -- python3 tools/bench_kernels.py --post --plain > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcW -- python3 tools/bench_kernels.py --post --plain > /dev/null 2>&1 &&
python3 tools/pmc_traffic.py $O/pmcF $O/pmcW 4096 > $O/post_step_pmc_traffic.json &&
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $O/pmcA -- python3 tools/bench_kernels.py --post --plain > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmcB -- python3 tools/bench_kernels.py --post --plain > /dev/null 2>&1 &&
python3 tools/pmc_summary.py $O/pmcA track_post > $O/post_step_sq_counters_a.json &&
python3 tools/pmc_summary.py $O/pmcB track_post > $O/post_step_sq_counters_b.json
rc=$?
rm -rf $O/bench $O/post $O/pmcF $O/pmcW $O/pmcA $O/pmcB
tail -n 1 $O/bench_n1.json.log | cut -c1-400
ls -la $O
exit $rc
