# Dev: A/B of the image-side kernels (cin = 3 / cout = 3 layers) between variants/lib_base.so and the working tree's library
for dt in f32 bf16; do
for v in base new; do
  echo "== $dt $v"
  if [ $v = base ]; then export CN_LIB=/root/repo/variants/lib_base.so; else unset CN_LIB; fi
  CONV_SHAPES_KIND=fwd python scripts/conv_shapes_bench.py 16 $dt 2>/dev/null | grep -E "total|   27 | 147 " 
done; done
