#!/bin/bash
# Dev: ablation builds of the F(4x4) kernel into variants/lib_w4_ab<N>.so (N = bit mask, see winograd4.hip: W4_ABLATE)
set -e
cd /root/repo/confignet_amd/csrc
mkdir -p /root/repo/variants
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DW4_ABLATE=$n -c winograd4.hip -o /tmp/w4_ab$n.o &
done; wait
for n in "$@"; do
  objs=$(ls *.o | grep -v '^winograd4.o$' | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/variants/lib_w4_ab$n.so $objs /tmp/w4_ab$n.o
done
ls -la /root/repo/variants
