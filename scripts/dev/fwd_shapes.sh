export CN_NO_WINOGRAD=1
for shape in "fwd 16 32 32 192 384 3 2" "dgrad 16 32 32 192 384 3 2" "fwd 16 16 16 384 512 3 2" "dgrad 16 16 16 384 512 3 2" "fwd 16 16 16 256 256 3 1" "fwd 16 64 64 96 192 3 2" "dgrad 16 64 64 96 192 3 2" "fwd 8 16 16 1024 256 1 1" "fwd 16 64 64 256 256 3 1" "fwd 16 128 128 48 96 3 2" "dgrad 80 64 64 96 192 3 2" "dgrad 80 128 128 48 96 3 2" "fwd 8 64 64 64 32 4 1" "fwd 8 32 32 256 64 4 1" "fwd 16 8 8 512 512 3 1" "fwd 8 64 64 64 256 1 1"; do
  python scripts/conv_one.py $shape 30 2>/dev/null | grep -v amdgpu
done
