for shape in "wgrad 16 128 128 48 96 3 2" "wgrad 80 128 128 48 96 3 2" "wgrad 16 64 64 96 192 3 2" "wgrad 16 32 32 192 384 3 2" "wgrad 16 16 16 384 512 3 2" "wgrad 8 64 64 256 256 3 1" "wgrad 16 16 16 256 256 3 1" "wgrad 8 32 32 256 64 4 1"; do
  python scripts/conv_one.py $shape 30 2>/dev/null | grep -v amdgpu
done
