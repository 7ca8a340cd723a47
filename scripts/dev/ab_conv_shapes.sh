#!/bin/bash
# per-shape conv table with two builds of the library on one box: bash scripts/dev/ab_conv_shapes.sh OUT BASE.so
out=gpurun_out/$1; mkdir -p $out
CN_LIB=$PWD/$2 python scripts/conv_shapes_bench.py 16 f32 > $out/base.txt 2>/dev/null
python scripts/conv_shapes_bench.py 16 f32 > $out/new.txt 2>/dev/null
CN_LIB=$PWD/$2 python scripts/conv_shapes_bench.py 16 f32 > $out/base2.txt 2>/dev/null
python scripts/conv_shapes_bench.py 16 f32 > $out/new2.txt 2>/dev/null
head -1 $out/base.txt $out/new.txt $out/base2.txt $out/new2.txt
