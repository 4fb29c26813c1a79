#!/bin/bash
# VALU / total instruction counts of track_post_kernel per role ablation (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ab in 0x70000 0x30000 0x50000 0x60000 0x0; do
  rm -rf gpurun_out/pmcR
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcR -- python3 tools/bench_kernels.py --post --plain --ablate=$ab > /dev/null 2>&1
  echo "ablate=$ab (0x70000 none, 0x30000 only char, 0x50000 only ref, 0x60000 only tar, 0 full)"
  python3 tools/pmc_summary.py gpurun_out/pmcR track_post | python3 -c "
import json,sys
d=json.load(sys.stdin); print({k:round(v['mean']) for k,v in d.items()})"
done
rm -rf gpurun_out/pmcR
