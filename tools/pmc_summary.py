#!/usr/bin/env python3
"""Sum rocprofv3 --pmc CSV output per kernel and counter.  usage: tools/pmc_summary.py <dir> [...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k].add(row["Dispatch_Id"])
    for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0)):
        print("%s  (dispatches %d)" % (k, len(calls[k])))
        for c, v in sorted(tot[k].items()):
            print("    %-36s %.6g" % (c, v))


if __name__ == "__main__":
    main()
