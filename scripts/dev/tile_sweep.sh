export CN_NO_WINOGRAD=1
for cfg in 0 1 2; do
  for shape in "16 64 64 256 256 3 1" "4 64 64 256 256 3 1" "1 64 64 256 256 3 1" "16 32 32 192 384 3 2" "96 32 32 192 384 3 2"; do
    echo -n "cfg $cfg: "; CN_CFG=$cfg python scripts/conv_one.py fwd $shape 30
  done
done
