#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c2; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv" > $O/ops_conv.txt 2>&1; tail -15 $O/ops_conv.txt
timeout 900 python scripts/wgrad_bench.py 16 sweep > $O/wgrad_bench.txt 2> $O/wgrad_bench.err; tail -3 $O/wgrad_bench.err
head -60 $O/wgrad_bench.txt
