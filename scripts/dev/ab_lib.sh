#!/bin/bash
# A/B of two builds of the library on the same box: bash scripts/dev/ab_lib.sh OUTDIR BASE.so [steps]   (b = working tree's library)
out=gpurun_out/$1; base=$2; steps=${3:-30}
mkdir -p $out
for r in 1 2 3; do
  CN_LIB=$PWD/$base python bench.py --steps $steps --warmup 5 --no-cpu-baseline > $out/a$r.json 2>> $out/err.txt
  python bench.py --steps $steps --warmup 5 --no-cpu-baseline > $out/b$r.json 2>> $out/err.txt
done
python - $out <<'PY'
import json, sys, glob
for tag in "ab":
    for f in sorted(glob.glob(sys.argv[1] + "/%s?.json" % tag)):
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1])
            print(tag, j["value"], j["ms_per_step"])
        except Exception as e:
            print(tag, f, "unreadable", e)
PY
