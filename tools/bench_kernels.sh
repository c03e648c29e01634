#!/bin/bash
# usage: tools/bench_kernels.sh <tag> [bench args]: one C2 bench run, per-kernel ms of the step on one line
tag=$1; shift
python bench.py --steps 3 --warmup 1 --no-cpu "$@" > gpurun_out/bk_$tag.json 2> gpurun_out/bk_$tag.err || exit 1
python - <<PY
import json
d=json.load(open("gpurun_out/bk_$tag.json"))
print("$tag", "ms/step %.2f" % d["ms_per_step"], [(k["name"][:30], round(k["ms_per_step"],2)) for k in d["roofline"]["kernels"]])
PY
