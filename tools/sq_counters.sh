#!/bin/bash
# usage: tools/sq_counters.sh <tag> [bench args] -- SQ / LDS / L2 counters of one bench step per kernel, in separate passes (8 SQ
# slots, 4 TCC slots per pass; --kernel-trace only beside --pmc), summarised as profiles-style JSON in gpurun_out/sq_<tag>.json
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pass() { # name counters...
  n=$1; shift
  if [ -n "$SQ_PROGRAM" ]; then      # another tool instead of bench.py: SQ_PROGRAM=tools/c4_check.py tools/sq_counters.sh c4 50000000 51 150 3
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/sq_${tag}_$n -- python3 $R/$SQ_PROGRAM $BENCH_ARGS > $R/gpurun_out/sq_${tag}_$n.log 2>&1 || tail -3 $R/gpurun_out/sq_${tag}_$n.log
    return
  fi
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/sq_${tag}_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-h2d $BENCH_ARGS > $R/gpurun_out/sq_${tag}_$n.log 2>&1 || tail -3 $R/gpurun_out/sq_${tag}_$n.log
}
BENCH_ARGS="$*"
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
pass d TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.Counter()
for n in "abcd":
    f = glob.glob("$R/gpurun_out/sq_${tag}_%s/*/*counter_collection.csv" % n)
    if not f: print("no counter file for pass", n); continue
    seen = set()
    for r in csv.DictReader(open(f[0])):
        if "kmr::" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if n == "a" and (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); launches[k] += 1
out = {}
for k, d in agg.items():
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc < 1e6: continue
    e = {"launches": launches[k]}
    e.update({c: v for c, v in sorted(d.items())})
    if wc:
        e["share_of_wave_cycles"] = {"issuing (ACTIVE_INST_ANY)": d.get("SQ_ACTIVE_INST_ANY", 0) / wc, "parked on s_waitcnt / barrier (WAIT_ANY)": d.get("SQ_WAIT_ANY", 0) / wc,
                                     "issue stall (WAIT_INST_ANY)": d.get("SQ_WAIT_INST_ANY", 0) / wc, "VALU": d.get("SQ_ACTIVE_INST_VALU", 0) / wc, "LDS": d.get("SQ_ACTIVE_INST_LDS", 0) / wc}
    if d.get("SQ_BUSY_CYCLES"): e["mean_waves_in_flight_per_SQ_busy_cycle"] = wc / d["SQ_BUSY_CYCLES"]
    if d.get("SQ_LDS_IDX_ACTIVE"): e["lds_bank_conflict_share"] = d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"]
    if d.get("TCC_REQ_sum"): e["l2_hit_rate"] = d.get("TCC_HIT_sum", 0) / d["TCC_REQ_sum"]
    out[k] = e
json.dump(out, open("$R/gpurun_out/sq_${tag}.json", "w"), indent=1)
for k, e in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    print(k[:60], {a: round(b, 3) for a, b in e.get("share_of_wave_cycles", {}).items()}, "waves/SQ %.2f" % e.get("mean_waves_in_flight_per_SQ_busy_cycle", 0), "L2 hit %.2f" % e.get("l2_hit_rate", -1), "VALU/SALU/LDS insts %.3g %.3g %.3g" % (e.get("SQ_INSTS_VALU", 0), e.get("SQ_INSTS_SALU", 0), e.get("SQ_INSTS_LDS", 0)))
PY
