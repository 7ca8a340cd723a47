#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4c9; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -s -m gpu -k "f4x4" 2>&1 | grep -E "F\(4x4\)|passed|failed|Error|assert|err" > $O/tests.txt; tail -15 $O/tests.txt
timeout 300 python scripts/dev/wino_bench.py > $O/wino_bench.txt 2>&1; cat $O/wino_bench.txt | tail -8
