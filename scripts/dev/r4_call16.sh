#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4c16; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py -x -q -m gpu -k "rotate or generator" > $O/tests.txt 2>&1; tail -4 $O/tests.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["step_functions_ms"])
PY
