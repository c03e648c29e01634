#!/usr/bin/env python3
"""print the per-kernel breakdown of a bench.py JSON line (development aid)"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ev = d["roofline"].get("hip_event_ms_per_step", {})
print("%.2f ms/step  %.2f G k-mers/s  frac %.4f  build %.2f  finalize %.2f" % (d["ms_per_step"], d["value"] / 1e9, d["roofline"]["frac"], ev.get("build", 0), ev.get("finalize", 0)))
for k in d["roofline"].get("kernels", []):
    print("  %-74s %6.2f ms  %.3f" % (k["name"][:74], k["ms_per_step"], k["frac"]))
if "value_incl_h2d" in d:
    print("  incl. H2D: %.2f ms/step  %.2f G k-mers/s  (copy alone %.1f ms, %.1f GB/s)" % (d["h2d"]["ms_per_step_incl_h2d"], d["value_incl_h2d"] / 1e9, d["h2d_ms"], d["h2d"]["h2d_GBps"]))
    if "as_text" in d["h2d"]:
        t = d["h2d"]["as_text"]
        print("  incl. H2D as text: %.2f ms/step  %.2f G k-mers/s  (copy alone %.1f ms, %.1f GB/s)" % (t["ms_per_step_incl_h2d"], t["value_incl_h2d"] / 1e9, t["h2d_ms"], t["h2d_GBps"]))
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("  cpu: %.2f M k-mers/s on %d threads (%.1f s)" % (c["value"] / 1e6, c["cores"], c["seconds"]))
if "exchange" in d:
    print("  exchange:", d["exchange"])
