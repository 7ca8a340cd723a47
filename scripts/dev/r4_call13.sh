#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4c13; mkdir -p $O
timeout 1800 python -m pytest tests/test_bf16_gpu.py tests/test_golden.py tests/test_metrics_gpu.py tests/test_nets_gpu.py tests/test_ops_gpu.py tests/test_steps_gpu.py -q -m gpu > $O/full_suite.txt 2>&1; tail -6 $O/full_suite.txt
