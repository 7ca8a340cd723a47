#!/bin/bash
# Dev: ablation builds of the LDS-DMA forward loop into variants/lib_fwd2_ab<N>.so (N = bit mask, see fwd2.hip: FWD2_ABLATE)
set -e
cd /root/repo/confignet_amd/csrc
mkdir -p /root/repo/variants
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DFWD2_ABLATE=$n -c fwd2.hip -o /tmp/fwd2_ab$n.o &
done; wait
for n in "$@"; do
  objs=$(ls *.o | grep -v '^fwd2.o$' | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/variants/lib_fwd2_ab$n.so $objs /tmp/fwd2_ab$n.o
done
ls -la /root/repo/variants
