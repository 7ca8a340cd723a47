#!/bin/bash
# Dev: build the library of another git revision of csrc/ next to the working tree's: scripts/dev/build_variant.sh <rev> <out.so>
set -e
REV=${1:-HEAD}; OUT=${2:-/root/repo/gpurun_out/lib_base.so}   # (gpurun_out/ does not travel to the GPU box: pass a path under the tree, e.g. variants/lib_base.so, for a same-box A/B -- and delete it afterwards)
T=$(mktemp -d); mkdir -p $T/confignet_amd/csrc $T/include
for f in $(git ls-tree --name-only $REV confignet_amd/csrc/); do git show $REV:$f > $T/$f; done
git show $REV:include/confignet_hip.h > $T/include/confignet_hip.h
cd $T/confignet_amd/csrc
for s in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -c $s -o ${s%.hip}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT *.o
echo $OUT
