#!/bin/bash
# usage: tools/ab.sh <variant> ... -- the C2 bench with each of kmernator_amd/csrc/build/v/<variant>.so in turn, on one box (development aid:
# boxes of the pool differ by a few per cent, variants are only comparable within one call); repeats the list twice
L=$GRAFT_REPO_ROOT/kmernator_amd/csrc
cp $L/libkmernator_amd.so /tmp/orig.so
for rep in 1 2; do
for v in "$@"; do
  cp $L/build/v/$v.so $L/libkmernator_amd.so
  timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu --no-h2d $AB_ARGS > $GRAFT_REPO_ROOT/gpurun_out/ab_$v.json 2> $GRAFT_REPO_ROOT/gpurun_out/ab_$v.err || { echo "$v failed"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/ab_$v.err; }
  echo "== $v"; python3 $GRAFT_REPO_ROOT/tools/kern.py $GRAFT_REPO_ROOT/gpurun_out/ab_$v.json | head -4
done; done
cp /tmp/orig.so $L/libkmernator_amd.so
