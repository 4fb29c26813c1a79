#!/bin/bash
# SQ counters of sim_step_bpl_kernel (run on the GPU box): two passes of 8 counters, product library, the product kernel alone
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcS1 gpurun_out/pmcS2
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmcS1 -- python3 tools/bench_sim.py --plain > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/pmcS2 -- python3 tools/bench_sim.py --plain > /dev/null 2>&1 &&
python3 - <<'PY'
import json, subprocess
out = {}
for d in ("gpurun_out/pmcS1", "gpurun_out/pmcS2"):
    out.update(json.loads(subprocess.check_output(["python3", "tools/pmc_summary.py", d, "sim_step_bpl", "--expect-waves=1024"] if d.endswith("1") else ["python3", "tools/pmc_summary.py", d, "sim_step_bpl"])))
out["note"] = ("rocprofv3 --kernel-trace --pmc <8 counters> (two passes) -- python3 tools/bench_sim.py --plain; sim_step_bpl_kernel, 4096 envs = 1024 "
               "one-wave workgroups, 4 substeps per launch; per-dispatch means")
json.dump(out, open("gpurun_out/sim_step_sq_counters.json", "w"), indent=1)
print({k: round(v["mean"]) for k, v in out.items() if isinstance(v, dict)})
PY
rm -rf gpurun_out/pmcS1 gpurun_out/pmcS2
