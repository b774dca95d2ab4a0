"""Developer helper: one-line digest of a bench.py JSON line read from stdin."""
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
def num(x):
    return x.get("value") if isinstance(x, dict) else x
r = j.get("roofline", {})
print(sys.argv[1] if len(sys.argv) > 1 else "bench", "value", round(j["value"], 1), "steady", num(j.get("value_steady")), "frac", round(r.get("frac", 0), 4),
      "launch us", round(r.get("avg_launch_us", 0), 1), "host", num(j.get("value_host_input")))
