#!/usr/bin/env python3
"""Short view of a bench.py JSON line:  tools/show_bench_line.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline %.0f Mpaths/s, %.1f ms/step, frac %.3f, byte ratio %.2f, n_gpus %d, scaling %s" %
      (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("hbm_algorithmic_ratio_to_hbm_peak", 0), d["n_gpus"], d["scaling"]))
print("data:", d["data"][:110])
for k, v in d.get("workloads", {}).items():
    rf = v.get("roofline")
    print("  %-14s %.0f Mpaths/s, %.1f ms/step%s%s" % (k, v["value"], v["ms_per_step"], (", frac %.3f" % rf["frac"]) if rf else "",
          (", render %.1f ms reduce %.2f ms (%s, %d ranks)" % (v["render_ms"], v["reduce_ms"], v["scaling"], v["rccl_ranks"])) if "render_ms" in v else ""))
for k, v in d.get("fast_modes", {}).items():
    print("  fast %-10s" % k, {a: (round(b["value"]), round(b["speedup_over_reference_walk"], 3)) for a, b in v.items()})
cb = d.get("cpu_baseline")
if cb:
    print("cpu baseline: %.3f Mpaths/s sample (%d threads), cfg1 whole %.2f Mpaths/s" % (cb["value"], cb["threads"], cb["cfg1"]["value"]))
