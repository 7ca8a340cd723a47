#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c4; mkdir -p $O
timeout 1500 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu -k "data_parallel" > $O/bf16dp.txt 2>&1; tail -25 $O/bf16dp.txt
