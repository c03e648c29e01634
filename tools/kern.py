#!/usr/bin/env python3
"""print the per-kernel breakdown of a bench.py JSON line (development aid)"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%.2f ms/step  %.2f G k-mers/s  frac %.4f  build %.2f  finalize %.2f" % (d["ms_per_step"], d["value"] / 1e9, d["roofline"]["frac"], d["roofline"]["build_ms_per_step"], d["roofline"]["finalize_ms_per_step"]))
for k in d["roofline"].get("kernels", []):
    print("  %-58s %6.2f ms  %.3f" % (k["name"], k["ms_per_step"], k["frac"]))
