#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4c10; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py -x -q -m gpu -k "conv or winograd or perceptual or second_stage" > $O/tests.txt 2>&1; tail -4 $O/tests.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench_w4.json 2> $O/bench.err
CN_NO_WINO4=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_now4.json 2>> $O/bench.err
for f in w4 now4; do python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", d["value"], d["ms_per_step"], d["step_functions_ms"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], d.get("loss_parity_vs_cpu"))
except Exception as e: print("$f failed", e)
PY
done
tail -3 $O/bench.err
