#!/bin/bash
# usage: tools/list_aim_sweep.sh "<aims>" [bench args] -- the C2 bench with kmr_tune list_aim = k-mers per super-k-mer list set to each value in
# turn (one box, one call: boxes differ by a few per cent), per-kernel times of each; 0 = the library's own choice
R=$GRAFT_REPO_ROOT
aims=$1; shift
for rep in 1 2; do
for a in $aims; do
  t=""; [ "$a" != "0" ] && t="--tune list_aim=$a"
  timeout -k 10 200 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu --no-h2d $t "$@" > $R/gpurun_out/aim_$a.json 2> $R/gpurun_out/aim_$a.err || { echo "aim $a failed"; tail -3 $R/gpurun_out/aim_$a.err; }
  echo "== list_aim $a"; python3 $R/tools/kern.py $R/gpurun_out/aim_$a.json | head -4
done; done
