#!/bin/bash
# usage: tools/c4_ab.sh <variant> ... -- C4 (tools/c4_check.py 50000000 51 150 3) with each of kmernator_amd/csrc/build/v/<variant>.so in turn on one box, twice
# (tools/variant.sh builds them; development aid).  Round 4: the count pass of two-word keys with -sink-insts-to-avoid-spills + max-ilp 83.2 / 84.2 ms,
# sink alone 83.3 / 82.8, max-ilp alone 97.1 / 98.0, neither 105.6 / 104.7.
L=$GRAFT_REPO_ROOT/kmernator_amd/csrc
cp $L/libkmernator_amd.so /tmp/orig.so
for rep in 1 2; do
for v in "$@"; do
  cp $L/build/v/$v.so $L/libkmernator_amd.so
  timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/c4_check.py 50000000 51 150 3 > $GRAFT_REPO_ROOT/gpurun_out/c4ab_$v.log 2>&1 || { echo "$v failed"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/c4ab_$v.log; }
  echo "== $v"; grep "rep 1\|digest" $GRAFT_REPO_ROOT/gpurun_out/c4ab_$v.log | cut -c1-150
done; done
cp /tmp/orig.so $L/libkmernator_amd.so
