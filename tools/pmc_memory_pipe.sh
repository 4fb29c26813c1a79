#!/bin/bash
# Where track_post_kernel's time goes below the instruction counts: issue-active and wait cycles of the SQ, texture-addresser (TA), L1 (TCP)
# and texture-data (TD) busy / stall cycles, L1->L2 requests.  Separate --pmc passes (one counter group each), plain kernel loop.
# usage (GPU box): bash tools/pmc_memory_pipe.sh [ablate-mask]    -> one JSON line per pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
AB=${1:-0x0}
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES TA_TOTAL_WAVEFRONTS TA_FLAT_READ_WAVEFRONTS TA_FLAT_WRITE_WAVEFRONTS" \
           "TCP_TOTAL_ACCESSES TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_GATE_EN1 TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES" \
           "TD_TD_BUSY TD_TC_STALL TD_LOAD_WAVEFRONT TD_STORE_WAVEFRONT TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY TCP_TOTAL_READ TCP_TOTAL_WRITE"; do
  rm -rf gpurun_out/pmcM
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcM -- python3 tools/bench_kernels.py --post --plain --ablate=$AB > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmcM track_post | python3 -c "
import json,sys
d=json.load(sys.stdin); print(json.dumps({k:round(v['mean']) for k,v in d.items()}))"
done
rm -rf gpurun_out/pmcM
