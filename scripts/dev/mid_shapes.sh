# representative mid-size layers of the 64x64-tile class (forward / data gradient of the discriminator blocks, ResNet stage 3)
for shape in "fwd 16 32 32 192 384 3 2" "dgrad 16 32 32 192 384 3 2" "fwd 16 16 16 384 512 3 2" "dgrad 16 16 16 384 512 3 2" "fwd 16 16 16 256 256 3 1" "fwd 16 64 64 96 192 3 2" "dgrad 16 64 64 96 192 3 2" "fwd 8 16 16 1024 256 1 1"; do
  CN_NO_WINOGRAD=1 python scripts/conv_one.py $shape 50 2>/dev/null | grep -v amdgpu
done
