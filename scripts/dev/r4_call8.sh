#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4c8; mkdir -p $O
timeout 300 python bench.py --no-cpu-baseline > $O/bench_bal.json 2> $O/bench.err
CN_NO_WGRAD_BALANCE=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_nobal.json 2>> $O/bench.err
CN_WGRAD_GROUP=4 timeout 300 python bench.py --no-cpu-baseline > $O/bench_bal4.json 2>> $O/bench.err
CN_WGRAD_GROUP=16 timeout 300 python bench.py --no-cpu-baseline > $O/bench_bal16.json 2>> $O/bench.err
for f in bal nobal bal4 bal16; do python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", d["value"], d["ms_per_step"], d["step_functions_ms"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"])
except Exception as e: print("$f failed", e)
PY
done
tail -5 $O/bench.err
timeout 1200 python -m pytest tests/test_steps_gpu.py -x -q -m gpu > $O/steps.txt 2>&1; tail -5 $O/steps.txt
timeout 200 python scripts/predict_latency.py > $O/predict_default.txt 2>/dev/null
CN_NO_IGEMM_ROWS=1 timeout 200 python scripts/predict_latency.py > $O/predict_norows.txt 2>/dev/null
CN_NO_GEMM1X1=1 timeout 200 python scripts/predict_latency.py > $O/predict_nog1.txt 2>/dev/null
grep "N=1 (replayed" $O/predict_*.txt
