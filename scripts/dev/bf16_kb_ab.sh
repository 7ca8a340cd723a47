# Dev: bf16 forward / data-gradient launches with 16- vs 32-deep stages of the LDS-DMA loop (CN_FWD2_BF16_KB; the rule in cn_fwd2_bf16)
for v in 16 32; do
  for k in fwd dgrad; do
  echo "== KB $v $k"
  CN_FWD2_BF16_KB=$v CONV_SHAPES_KIND=$k python scripts/conv_shapes_bench.py 16 bf16 2>/dev/null
  done
done
