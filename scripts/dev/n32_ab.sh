# Dev: A/B of the 128 x 32 tile of the LDS-DMA loop (CN_FWD2_N32) on the iteration's 32-output-channel launches
for v in 0 1; do
  echo "== CN_FWD2_N32=$v"
  CN_FWD2_N32=$v python scripts/conv_shapes_bench.py 16 f32 2>/dev/null | grep -E "total|^(fwd|dgrad) .*    32  nd"
done
