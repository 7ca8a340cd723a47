#!/bin/bash
# A/B of one environment switch on the same box: bash scripts/dev/ab_env.sh OUTDIR VAR [steps]
# runs bench.py (no CPU baseline) alternately with VAR unset / VAR=1, three times each.
out=gpurun_out/$1; var=$2; steps=${3:-30}
mkdir -p $out
for r in 1 2 3; do
  python bench.py --steps $steps --warmup 5 --no-cpu-baseline > $out/a$r.json 2>> $out/err.txt
  env $var=1 python bench.py --steps $steps --warmup 5 --no-cpu-baseline > $out/b$r.json 2>> $out/err.txt
done
python - $out <<'PY'
import json, sys, glob
for tag in "ab":
    for f in sorted(glob.glob(sys.argv[1] + "/%s?.json" % tag)):
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1])
            print(tag, j["value"], j["ms_per_step"], j.get("phases_ms"))
        except Exception as e:
            print(tag, f, "unreadable", e)
PY
