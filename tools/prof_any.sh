#!/bin/bash
# usage: tools/prof_any.sh <tag> <script.py> [args]  -- rocprofv3 kernel stats of any python tool into gpurun_out/prof_<tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
python3 -c "
import csv,glob
rows=list(csv.DictReader(open(glob.glob('$GRAFT_REPO_ROOT/gpurun_out/prof_$tag/*/*kernel_stats.csv')[0])))
for r in rows[:24]: print('%-70s calls %4s avg %9.3f ms total %8.1f ms' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6))
"
