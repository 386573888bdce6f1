"""Reads bench.py's JSON line on stdin and prints the fields an A/B needs on one line:  python bench.py ... | python profiles/line_fields.py <label>"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1] if len(sys.argv) > 1 else "-", "value %.1f (%.1f..%.1f) ms/step %.4f avg_launch_ms %.5f frac %.4f" % (
    d["value"], d.get("value_min", 0), d.get("value_max", 0), d["ms_per_step"], r.get("avg_launch_ms", 0), r["frac"]), flush=True)
