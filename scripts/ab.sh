#!/bin/bash
# A/B of environment switches on the benchmark's timed loop, same box, alternating: scripts/ab.sh "CN_X=1" "CN_Y=1" ...
# (first column: the variant; "base" = no switch).  Prints images/s and ms per iteration of 3 rounds each.
export CN_BENCH_SKIP_ROOFLINE_PASS=1
for round in 1 2 3; do
  for v in base "$@"; do
    if [ "$v" = base ]; then e=""; else e="$v"; fi
    r=$(env $e python bench.py --no-cpu-baseline --steps ${AB_STEPS:-20} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_functions_ms'])")
    echo "$v: $r"
  done
done
