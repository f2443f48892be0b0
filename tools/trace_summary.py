#!/usr/bin/env python3
"""Per-kernel duration statistics from a rocprofv3 kernel trace of bench.py.

    tools/trace_summary.py <kernel_trace.csv> [--last-launches N]

In a back-to-back stream the trace stamps a dispatch's start when the command processor picks the
packet up, which can be before the previous kernel of the same stream has drained, so summed
durations exceed wall time.  bench.py's last-but-one step brackets every launch with HIP events,
which separates the dispatches; `--last-launches N` restricts a kernel's statistics to its last N
launches of a given template instance so that step can be read on its own.
"""
import argparse
import csv
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--last-launches", type=int, default=0)
    a = ap.parse_args()
    d = defaultdict(list)
    for r in csv.DictReader(open(a.trace)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    print("kernel,calls,total_ms,avg_us,min_us,max_us")
    for name, v in sorted(d.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
        v.sort()
        durs = [x[1] for x in v]
        if a.last_launches:
            durs = durs[-a.last_launches:]
        print("%s,%d,%.3f,%.1f,%.1f,%.1f" % (name, len(durs), sum(durs) / 1e6, sum(durs) / len(durs) / 1e3, min(durs) / 1e3, max(durs) / 1e3))


if __name__ == "__main__":
    main()
