#!/bin/bash
# SQ counters of track_post_kernel (two rocprofv3 --pmc passes); run on the GPU box: bash tools/pmc_post.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmcA -- python3 tools/bench_kernels.py --post --plain > gpurun_out/pmcA.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmcB -- python3 tools/bench_kernels.py --post --plain > gpurun_out/pmcB.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmcC -- python3 tools/bench_kernels.py --post --plain > gpurun_out/pmcC.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmcA track_post > gpurun_out/pmcA.json
python3 tools/pmc_summary.py gpurun_out/pmcB track_post > gpurun_out/pmcB.json
python3 tools/pmc_summary.py gpurun_out/pmcC track_post > gpurun_out/pmcC.json
python3 -c "
import json
for f in ('gpurun_out/pmcA.json','gpurun_out/pmcB.json','gpurun_out/pmcC.json'):
    d=json.load(open(f)); print({k:round(v['mean']) for k,v in d.items()})
"
tail -n 2 gpurun_out/pmcC.log
rm -rf gpurun_out/pmcA gpurun_out/pmcB gpurun_out/pmcC
