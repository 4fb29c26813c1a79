#!/bin/bash
# Where track_post_kernel's time goes below the instruction counts: issue-active and wait cycles of the SQ.  Separate --pmc passes (one
# counter group each), plain kernel loop.  The texture-addresser / L1 / texture-data groups - TA_*, TCP_*, TD_* - are NOT collected:
# in round 3 ONE pass asked for eight TA_* counters at once (TA_TA_BUSY ... TA_FLAT_WRITE_WAVEFRONTS; TA is instanced per CU, 256 of them
# over 8 XCDs), rocprofv3 wrote nothing - neither its banner nor an error - and the call was killed after 7 silent minutes.  What is known:
# the SQ_* passes of the very same command line and kernel return in seconds, so it is the counter request, not the kernel; whether it
# is the group's size or a TA counter this rocprofv3 cannot program on gfx950 was not determined, because that would take re-running a
# pass that hangs a box.  If TA / TCP evidence is ever needed: at most two counters of one block per pass, under `gpurun --timeout 120`,
# once.  Nothing in rounds 3-4 needed it (the kernels it would explain are issue-bound, profiles/r04_*_sq_counters.json).
# usage (GPU box): bash tools/pmc_memory_pipe.sh [ablate-mask]    -> one JSON line per pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
AB=${1:-0x0}
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"; do
  rm -rf gpurun_out/pmcM
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcM -- python3 tools/bench_kernels.py --post --plain --ablate=$AB > /dev/null 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmcM track_post | python3 -c "
import json,sys
d=json.load(sys.stdin); print(json.dumps({k:round(v['mean']) for k,v in d.items()}))"
done
rm -rf gpurun_out/pmcM
