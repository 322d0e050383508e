#!/bin/bash
# build a variant of the library with extra -D flags into othellozero_amd/lib_<name>/ (kernel experiments)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
out=othellozero_amd/lib_$name; mkdir -p $out
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wno-unused-value -Wno-unused-result"
for f in oz_rules oz_search oz_net oz_train; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c othellozero_amd/csrc/$f.hip -o $out/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libothellozero_amd.so $out/oz_rules.o $out/oz_search.o $out/oz_net.o $out/oz_train.o
echo built $out
