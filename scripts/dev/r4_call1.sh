#!/bin/bash
# round 4, call 1: forced-branch whole-step tests (x5 default, x3 deterministic), changed op tests, full suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c1; mkdir -p $O
for i in 1 2 3 4 5; do
  timeout 900 python -m pytest tests/test_nets_gpu.py -x -q -s -m gpu -k "first_stage_generator_step_and_adam or second_stage_generator_step" 2>&1 | grep -E "check_grads|passed|failed|Error|assert" >> $O/forced_default.txt
done
for i in 1 2 3; do
  CN_DETERMINISTIC=1 timeout 900 python -m pytest tests/test_nets_gpu.py -x -q -s -m gpu -k "first_stage_generator_step_and_adam or second_stage_generator_step" 2>&1 | grep -E "check_grads|passed|failed|Error|assert" >> $O/forced_det.txt
done
timeout 1500 python -m pytest tests -x -q -m gpu > $O/full_suite.txt 2>&1
tail -5 $O/full_suite.txt
cat $O/forced_default.txt | tail -30
cat $O/forced_det.txt | tail -12
