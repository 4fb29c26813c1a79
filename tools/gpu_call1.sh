#!/bin/bash
# round-4 GPU call 1: the whole gpu suite on the refactored libraries, then two probes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c1; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -n 5 $O/pytest.log
timeout -k 10 120 python3 tools/probe/event_in_graph.py > $O/event_in_graph.txt 2>&1; tail -n 8 $O/event_in_graph.txt
timeout -k 10 120 python3 tools/probe/graph_node_cost.py > $O/node_cost_plain.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nc -- python3 tools/probe/graph_node_cost.py > $O/node_cost_rocprof.txt 2>&1
find $O/nc -name "*kernel_stats.csv" -exec cp {} $O/node_cost_kernel_stats.csv \;
rm -rf $O/nc
cat $O/node_cost_plain.txt; head -n 14 $O/node_cost_kernel_stats.csv | cut -c1-200
