#!/usr/bin/env python3
"""HBM traffic per kernel launch from rocprofv3 FETCH_SIZE / WRITE_SIZE passes of bench.py.

    tools/make_traffic.py <profile-dir> <workload> [<out.json>]

<profile-dir> holds FETCH_SIZE/ and WRITE_SIZE/ as written by tools/profile_bench.sh.  Both counters
are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64 bytes,
so it is doubled; WRITE_SIZE is exact for streaming stores.  The result is the mean over all launches
of a kernel in the profiled run, in bytes, merged into <out.json> as {workload: {kernel: bytes}}.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(d, counter):
    tot, ids = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            tot[k] += float(row["Counter_Value"])
            ids[k].add(row["Dispatch_Id"])
    return tot, ids


def main():
    d, workload = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    fetch, fid = per_kernel(os.path.join(d, "FETCH_SIZE"), "FETCH_SIZE")
    write, wid = per_kernel(os.path.join(d, "WRITE_SIZE"), "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        nf, nw = max(len(fid[k]), 1), max(len(wid[k]), 1)
        rd = 2.0 * fetch.get(k, 0.0) * 1024.0 / nf
        wr = write.get(k, 0.0) * 1024.0 / nw
        res[k] = rd + wr
        print("%-16s launches %5d  read %12.0f B  write %12.0f B  per launch" % (k, nf, rd, wr))
    data = {}
    if os.path.exists(out):
        data = json.load(open(out))
    data[workload] = res
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
