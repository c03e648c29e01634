#!/bin/bash
# usage: tools/variant.sh <name> <tu> [-DFLAG ...] -- a variant of the library that differs in ONE instance translation unit
# (e.g. kmr_inst_skc1) compiled with extra flags: kmernator_amd/csrc/build/v/<name>.so (development aid for tools/ab.sh)
name=$1; tu=$2; shift 2
cd $(dirname $0)/../kmernator_amd/csrc && mkdir -p build/v
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -w"
/opt/rocm/bin/hipcc $FLAGS "$@" -c -o build/v/${name}_$tu.o $tu.hip || exit 1
OBJS=$(ls build/*.o | grep -v "/$tu.o")
/opt/rocm/bin/hipcc $FLAGS -shared -Wl,-z,defs -o build/v/$name.so $OBJS build/v/${name}_$tu.o -ldl && echo built build/v/$name.so
