export CONV_SHAPES_KIND=wgrad
for v in 384 512 768 1024 1536 2048; do
  echo "== $v"; CN_WG_BLOCKS=$v python scripts/conv_shapes_bench.py 16 f32 2>/dev/null | head -60
done
