#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c3; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py -x -q -m gpu -k "residual or real_encoder or second_stage" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err
for f in new; do python - <<PY
import json
d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], d["step_functions_ms"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"])
PY
done
tail -5 $O/bench_new.err
timeout 900 python -m pytest tests/test_steps_gpu.py -x -q -m gpu > $O/steps.txt 2>&1; tail -5 $O/steps.txt
