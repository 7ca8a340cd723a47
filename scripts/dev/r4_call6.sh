#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4c6; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py -x -q -m gpu -k "tail_statistics or stacked_generator or discriminator or second_stage" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench_red4.json 2> $O/bench.err
CN_NO_TAIL_STATS4=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_nored4.json 2>> $O/bench.err
for f in red4 nored4; do python - <<PY
import json
d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], d["step_functions_ms"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["launches_per_step"])
PY
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/gt -- python $GRAFT_REPO_ROOT/scripts/g_step_trace.py 6 > $O/g_step_line.txt 2>/dev/null
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(ls /tmp/gt/*/*.db | head -1) > $O/g_step_kernel_trace.txt
head -50 $O/g_step_kernel_trace.txt; cat $O/g_step_line.txt
