"""tools/bench_brief.py <bench json line file> -- the figures of a bench.py line one looks at first: headline, end to end, every leg."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("headline %.1f %s  frac %.4f  kernel %.4f ms  traffic_stale %s" % (d["value"], d["unit"], r["frac"], r.get("kernel_ms_per_launch", 0), r.get("traffic_stale")))
e = d.get("end_to_end", {})
print("end to end:", {k: e[k] for k in e if k.startswith("value") or "passes" in k})
for k, v in d.get("legs", {}).items():
    if not isinstance(v, dict):
        continue
    rf = v.get("roofline", {}) if isinstance(v.get("roofline"), dict) else {}
    e2e = v.get("end_to_end")
    print("  %-22s frac %s  ms %s  e2e %s %s" % (k, v.get("frac", rf.get("frac")), v.get("kernel_ms_per_launch", rf.get("kernel_ms_per_launch")),
                                               (e2e or {}).get("value") if isinstance(e2e, dict) else e2e, ("ERROR " + str(v["error"])) if "error" in v else ""))
