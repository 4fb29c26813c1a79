#!/bin/bash
# round-4 GPU call 2: gpu suite (device invariants of the simulator included) + a short training probe on the changed contact model
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c2; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/pytest.log
tail -n 25 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 420 python3 tools/train_probe.py --envs 4096 --workload flat_1clip --iters 1200 --max-seconds 330 --out $O/train_teaser.json > $O/train_teaser.log 2>&1
tail -n 3 $O/train_teaser.log
