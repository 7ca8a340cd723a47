# Dev: pipeline-level A/B of environment switches: alternating bench.py runs on one box (isolated per-shape timings of 20 us launches
# scatter too much to decide these).  usage: pipe_ab.sh <dtype> <reps> "ENV=1 ..." "ENV=2" ...
DT=$1; REPS=$2; shift 2
for i in $(seq $REPS); do for v in "$@"; do
  echo -n "$v: "
  env CN_BENCH_SKIP_ROOFLINE_PASS=1 $v python bench.py --dtype $DT --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"
done; done
