#!/bin/bash
# same-box A/B of the parity-ordered launches on the plain-GEMM loop vs igemm_fwd_kernel (CN_NO_IGEMM_ROWS_PAR=1)
for shape in "dgrad 16 128 128 48 96 3 2" "dgrad 16 64 64 96 192 3 2" "dgrad 16 32 32 192 384 3 2" "dgrad 16 16 16 384 768 3 2" "dgrad 96 16 16 384 768 3 2" "fwd 16 128 128 48 96 3 2" "fwd 16 64 64 96 192 3 2"; do
  for rep in 1 2; do
    a=$(python scripts/conv_one.py $shape 50 2>/dev/null | grep -v amdgpu | tail -1)
    b=$(CN_NO_IGEMM_ROWS_PAR=1 CN_NO_IGEMM_ROWS=1 python scripts/conv_one.py $shape 50 2>/dev/null | grep -v amdgpu | tail -1)
    echo "$shape | rows: $a | old: $b"
  done
done
