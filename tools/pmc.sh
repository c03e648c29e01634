#!/bin/bash
# HBM traffic counters of one bench step, one counter per pass (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-h2d "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$ctr.log 2>&1
  ls $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$ctr/*/ | head -5
done
python3 - <<PY
import csv, glob, collections
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_%s/*/*counter_collection.csv" % ctr)
    if not f: print("no counter file for", ctr); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        if "kmr::" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            k = r["Kernel_Name"][:50]; agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        print("%-10s %-50s launches %3d  total %12.0f (KB units -> %.2f GB)" % (ctr, k, n, v, v * 1024 / 1e9))
PY
