export CONV_SHAPES_KIND=wgrad
for v in "X=1" "CN_BF16_WGRAD_WGS=2048" "CN_BF16_WGRAD_ROWS=384" "CN_BF16_WGRAD_ROWS=768" "X=2"; do
  echo "== $v"; env $v python scripts/conv_shapes_bench.py 16 bf16 2>/dev/null | head -60
done
