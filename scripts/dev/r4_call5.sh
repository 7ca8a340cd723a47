#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c5; mkdir -p $O
timeout 900 python -m pytest tests/test_nets_gpu.py -x -q -m gpu -k "second_stage or full_iteration or hip_graph" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench_merge.json 2> $O/bench.err
CN_NO_G_MERGE=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_nomerge.json 2>> $O/bench.err
for f in merge nomerge; do python - <<PY
import json
d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], d["step_functions_ms"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["launches_per_step"])
PY
done
tail -5 $O/bench.err
timeout 1200 python -m pytest tests/test_steps_gpu.py -x -q -m gpu > $O/steps.txt 2>&1; tail -5 $O/steps.txt
