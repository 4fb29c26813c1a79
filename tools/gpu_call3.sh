#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c3; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/pytest.log
tail -n 30 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 1000 bash tools/profile_round.sh > $O/profile_round.log 2>&1; echo "profile rc=$?"; tail -n 20 $O/profile_round.log
