#!/usr/bin/env python3
"""tools/pmc_summary.py <tag> [quality] [commit] -- turn the two counter passes of tools/pmc.sh (gpurun_out/pmc_<tag>_FETCH_SIZE, ..._WRITE_SIZE)
into the per-step HBM traffic summary bench.py reads (profiles/rNN_pmc_traffic.json).  FETCH_SIZE is doubled (gfx950 reports half
the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section), WRITE_SIZE taken as reported, both in KB.  Every build of the
run handles the same reads, so a kernel's per-step figure is its total over the run divided by the number of builds (= launches of
the count kernel)."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
quality = sys.argv[2] if len(sys.argv) > 2 else "flat"
tot = {}
launches = collections.Counter()
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc_%s_%s/*/*counter_collection.csv" % (tag, ctr))
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f[0])):
        if "kmr::" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("kmr::", "")
            agg[name] += float(r["Counter_Value"]) * 1024
            if ctr == "FETCH_SIZE":
                launches[name] += 1
    tot[ctr] = agg
count_names = [n for n in launches if "count_kernel" in n and "unit" not in n and "owner" not in n]
builds = max(launches[n] for n in count_names)
per = {}
total = 0.0
for name in sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"]), key=lambda n: -(2 * tot["FETCH_SIZE"].get(n, 0) + tot["WRITE_SIZE"].get(n, 0))):
    fr = tot["FETCH_SIZE"].get(name, 0.0) / builds / 1e9
    wr = tot["WRITE_SIZE"].get(name, 0.0) / builds / 1e9
    if 2 * fr + wr < 0.005:
        continue
    per[name] = {"launches_per_step": round(launches[name] / builds, 2), "fetch_raw_GB": round(fr, 3), "fetch_corrected_GB": round(2 * fr, 3), "write_GB": round(wr, 3)}
    total += 2 * fr + wr
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of bench.py --steps 1 --warmup 0 --no-cpu (tools/pmc.sh %s): %d builds of the same batch in the run, figures are per build" % (tag, builds),
    "corrections": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; both x1024 (KB units)",
    "build_mode": "super-k-mer lists",
    "quality": quality,
    "reads": 10000000,
    "commit": sys.argv[3] if len(sys.argv) > 3 else "?",
    "per_step_GB": per,
    "hot_path_total_GB_per_step": round(total, 2),
}
json.dump(out, sys.stdout, indent=1)
print()
