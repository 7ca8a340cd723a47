#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4c12; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/full_suite.txt 2>&1; tail -4 $O/full_suite.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["step_functions_ms"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["launches_per_step"])
PY
timeout 300 python bench.py --no-cpu-baseline --dtype bf16 > $O/bench_bf16.json 2>> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_bf16.json").read().strip().splitlines()[-1])
print("bf16", d["value"], d["ms_per_step"], d["step_functions_ms"])
PY
