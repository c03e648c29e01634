#!/bin/bash
# usage: tools/pmc_any.sh <tag> "<CTR1 CTR2 ...>" [bench args] -- per-kernel sums of arbitrary PMC counters for one bench step
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log | cut -c1-300
python3 - <<PY
import csv, glob, collections
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/*/*counter_collection.csv")
if not f: print("no counter file"); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if "kmr::" in r["Kernel_Name"]:
        k = r["Kernel_Name"][:48]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k); print("    " + "  ".join("%s=%.4g" % kv for kv in sorted(d.items())))
PY
