#!/usr/bin/env python3
"""Timeline of one render from a rocprofv3 kernel trace: per dispatch of the pipeline's kernels its stream (queue),
start and end relative to the first, and how much of its span some other pipeline kernel was running too.

    tools/trace_timeline.py <kernel_trace.csv> [--from-ms A --to-ms B]
"""
import argparse
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--last", type=int, default=80, help="dispatches to show, counted from the end")
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.trace)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith("k_"):
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("<")[0], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    rows = rows[-a.last:]
    t0 = rows[0][0]
    for (s, e, name, q, st) in rows:
        other = 0
        for (s2, e2, n2, q2, st2) in rows:
            if (s2, e2, n2, q2) == (s, e, name, q):
                continue
            lo, hi = max(s, s2), min(e, e2)
            if hi > lo:
                other += hi - lo
        print("%-12s q%-3s st%-3s %9.3f -> %9.3f ms  (%7.3f ms, %5.1f%% beside another kernel)" %
              (name, q, st, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, 100.0 * other / max(e - s, 1)))
    print("span %.3f ms; sum of durations %.3f ms" % ((max(r[1] for r in rows) - t0) / 1e6, sum(r[1] - r[0] for r in rows) / 1e6))


if __name__ == "__main__":
    main()
