#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4c7; mkdir -p $O
timeout 900 python -m pytest tests/test_nets_gpu.py -x -q -s -m gpu -k "second_stage_generator_step" 2>&1 | grep -E "check_grads|passed|failed|Error" > $O/tests.txt; tail -6 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
for try in 1 2 3; do
  rm -rf /tmp/gt
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/gt -- python $GRAFT_REPO_ROOT/scripts/g_step_trace.py 6 > $O/g_step_line.txt 2>$O/g_step_err.txt
  if ls /tmp/gt/*/*.db > /dev/null 2>&1; then break; fi
  echo "no db (try $try)"; tail -3 $O/g_step_err.txt
done
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(ls /tmp/gt/*/*.db | head -1) > $O/g_step_kernel_trace.txt
head -64 $O/g_step_kernel_trace.txt; cat $O/g_step_line.txt | tail -2
