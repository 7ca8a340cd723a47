#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4c15; mkdir -p $O
timeout 600 python scripts/g_step_ops.py > $O/g_step_ops.txt 2>&1; cat $O/g_step_ops.txt | tail -140
