#!/bin/bash
# Dev: time the F(4x4) kernel's ablation builds (scripts/dev/wino4_ablate.sh) on the VGG shapes
echo "== full"; python scripts/dev/wino_bench.py 2>/dev/null | cut -c1-60
for n in 1 2 4 8 16 3 7 15 24; do
  echo "== ablate $n"; CN_LIB=/root/repo/variants/lib_w4_ab$n.so python scripts/dev/wino_bench.py 2>/dev/null | cut -c1-60
done
