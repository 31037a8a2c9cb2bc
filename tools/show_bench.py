#!/usr/bin/env python3
"""print the headline fields of a bench.py JSON line (file given as argv[1])"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value %.3f audio-h/s  %.1f ms/step  roofline %s frac %.4f  launch %.2f ms" % (d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["launch_ms"]))
print({k: v["ms"] for k, v in d["stages"].items()})
c = d.get("cpu_baseline")
if c:
    print("cpu 1 core: %.1f xRT (%s), agree %s" % (c["xRT"], c["sample"], c.get("one_best_agree")))
    if "n_core" in c: print("cpu n-core:", c["n_core"])
    if "config1" in c: print("cpu config1:", c["config1"])
