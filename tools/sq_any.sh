#!/bin/bash
# usage: tools/sq_any.sh <tag> <script.py> [args] -- the SQ counter passes of tools/sq_counters.sh for any python tool (per-kernel sums)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pass() { n=$1; shift; c="$1"; shift; rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/sqa_${tag}_$n -- python3 "$@" > $R/gpurun_out/sqa_${tag}_$n.log 2>&1 || tail -3 $R/gpurun_out/sqa_${tag}_$n.log; }
pass a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "$@"
pass b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "$@"
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for n in "ab":
    f = glob.glob("$R/gpurun_out/sqa_${tag}_%s/*/*counter_collection.csv" % n)
    if not f: print("no counter file", n); continue
    for r in csv.DictReader(open(f[0])):
        if "kmr::" in r["Kernel_Name"]: agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
    wc = d.get("SQ_WAVE_CYCLES", 1)
    print(k[:70]); print("   VALU %.3g SALU %.3g LDS %.3g | issuing %.2f waiting %.2f stall %.2f valu-share %.2f lds-share %.2f | lds conflict share %.2f" % (d["SQ_INSTS_VALU"], d["SQ_INSTS_SALU"], d["SQ_INSTS_LDS"], d["SQ_ACTIVE_INST_ANY"]/wc, d["SQ_WAIT_ANY"]/wc, d["SQ_WAIT_INST_ANY"]/wc, d["SQ_ACTIVE_INST_VALU"]/wc, d["SQ_ACTIVE_INST_LDS"]/wc, d["SQ_LDS_BANK_CONFLICT"]/max(1,d["SQ_LDS_IDX_ACTIVE"])))
PY
