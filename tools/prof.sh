#!/bin/bash
# usage: tools/prof.sh <tag> [bench args]  -- rocprofv3 kernel stats of bench.py into gpurun_out/prof_<tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-h2d "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
grep metric $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); h=d['roofline']['hip_event_ms_per_step']; print('%.2f G kmers/s  %.1f ms/step  build %.1f ms  finalize %.1f ms' % (d['value']/1e9, d['ms_per_step'], h['build'], h['finalize']))
except Exception as e: print('no json', e)"
python3 -c "
import csv,glob
rows=list(csv.DictReader(open(glob.glob('$GRAFT_REPO_ROOT/gpurun_out/prof_$tag/*/*kernel_stats.csv')[0])))
for r in rows[:10]: print('%-58s calls %4s avg %9.3f ms total %8.1f ms' % (r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e6, float(r['TotalDurationNs'])/1e6))
"
