"""Prints the few numbers of a bench.py JSON line that are compared between runs.  usage: show_bench.py bench.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("pairs/s %.0f  ms/step %.3f  kernel %s frac %.4f avg_launch_ms %.4f" % (d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["avg_launch_ms"]))
s = d.get("single_sequence") or {}
print("b1 ms/pair %s  b8 ms/step %s" % (s.get("b1", {}).get("ms_per_pair"), s.get("b8", {}).get("ms_per_step")))
print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], "parity mismatches", d["parity_check"]["mismatches"], "of", d["parity_check"]["pairs"])
print({k: round(v, 3) for k, v in d["kernels_ms_per_step_single_stream"].items() if v > 0.05})
lb = d.get("local_ba")
if lb:
    print("local_ba", {k: lb[k] for k in lb if k in ("ms_per_ba", "ms", "mode")})
