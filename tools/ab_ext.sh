#!/bin/bash
# usage: tools/ab_ext.sh <variant> ... -- tools/ext_check.py (mode 3 only) with each of kmernator_amd/csrc/build/v/<variant>.so, on one box
L=$GRAFT_REPO_ROOT/kmernator_amd/csrc
cp $L/libkmernator_amd.so /tmp/orig.so
for v in "$@"; do
  cp $L/build/v/$v.so $L/libkmernator_amd.so
  echo "== $v"; EXT_MODES=3 timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/ext_check.py ${EXT_N:-10000000} 21 ${EXT_Q:-flat} 2>&1 | grep "^mode\|groups"
done
cp /tmp/orig.so $L/libkmernator_amd.so
