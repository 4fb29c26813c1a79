#!/bin/bash
cd $GRAFT_REPO_ROOT
for opt in "-O2" "-O3 -fno-strict-aliasing" "-O3 -ffp-contract=off" "-O3 -fno-unroll-loops" "-O3 -fno-vectorize -fno-slp-vectorize" "-O3 -mllvm -amdgpu-promote-alloca-to-vector-limit=0" "-Os"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $opt -std=c++17 -fPIC -shared -o parc_amd/lib/libparc_hip.so parc_amd/csrc/parc_kin.hip parc_amd/csrc/parc_sim.hip 2>/dev/null
  echo "=== $opt"
  python tools/scratch/dbg_sim.py 2>&1 | grep -E "^dofs \[9\]|^all"
done
