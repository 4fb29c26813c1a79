#!/bin/bash
# SQ counters of sim_step_bpl_kernel (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcS
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmcS -- python3 tools/bench_sim.py > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmcS sim_step_bpl | python3 -c "
import json,sys
d=json.load(sys.stdin); print({k:round(v['mean']) for k,v in d.items()})"
rm -rf gpurun_out/pmcS
