export CN_NO_WINOGRAD=1
for shape in "fwd 16 32 32 192 384 3 2" "dgrad 16 32 32 192 384 3 2" "fwd 16 16 16 384 512 3 2" "fwd 16 16 16 256 256 3 1" "fwd 16 64 64 96 192 3 2" "fwd 8 16 16 1024 256 1 1" "fwd 16 64 64 256 256 3 1"; do
  for cfg in "" 2 5 6 7; do
    for sp in "" 1; do
      [ -z "$cfg" ] && [ -n "$sp" ] && continue
      echo -n "cfg=${cfg:-auto} splits=${sp:-auto}: "; CN_CFG=$cfg CN_SPLITS=$sp python scripts/conv_one.py $shape 30 2>/dev/null | grep -v amdgpu
    done
  done
done
