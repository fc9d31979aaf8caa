"""Prints the headline fields of the JSON line bench.py wrote into the given log file(s)."""
import json, sys
for f in sys.argv[1:]:
    l = [x for x in open(f) if x.startswith("{")][-1]
    d = json.loads(l)
    r = d.get("roofline", {})
    print(f, d["metric"], d["value"], d["ms_per_step"], "| roofline", r.get("ms_per_launch"), r.get("frac"))
