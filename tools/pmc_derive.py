#!/usr/bin/env python3
"""Derived utilisation figures per kernel from the counter passes of tools/pmc_passes.sh.

    tools/pmc_derive.py <dir with sq1/ sq2/ ta/ fetch/ write/ l2/>

SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count quad-cycles summed over waves; GRBM_GUI_ACTIVE
counts cycles per XCD (8 of them); FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE tallying 128-byte
requests at 64 bytes (doubled here).  256 CUs, 4 SIMDs each, 8 wave slots per SIMD.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    tot = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[k].add(row["Dispatch_Id"])
    for k in sorted(tot, key=lambda k: -tot[k].get("GRBM_GUI_ACTIVE", 0)):
        c = tot[k]
        if "GRBM_GUI_ACTIVE" not in c or "SQ_WAVE_CYCLES" not in c:
            continue
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0            # kernel cycles, summed over its dispatches
        simds = 256 * 4
        print("%s  (%d dispatches, %.2f ms at 2.4 GHz)" % (k, len(n[k]), cycles / 2.4e6))
        print("   wave occupancy            %5.1f %% of 8 slots per SIMD" % (100 * 4 * c["SQ_WAVE_CYCLES"] / (cycles * simds * 8)))
        print("   VALU busy                 %5.1f %% of SIMD cycles" % (100 * 4 * c["SQ_ACTIVE_INST_VALU"] / (cycles * simds)))
        if c.get("SQ_THREAD_CYCLES_VALU"):
            print("   lanes active in VALU ops  %5.1f %%" % (100 * c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"])))
        print("   wave time: issuing %4.1f %%, issue-stalled %4.1f %%, parked on s_waitcnt %4.1f %%" %
              (100 * c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
               100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
        if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
            print("   L2 hit rate               %5.1f %%" % (100 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            byts = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
            print("   HBM traffic               %5.2f GB, %6.0f GB/s = %4.1f %% of 8 TB/s" % (byts / 1e9, byts / (cycles / 2.4e9) / 1e9, 100 * byts / (cycles / 2.4e9) / 8e12))
        if c.get("SQ_INSTS_VALU"):
            print("   instructions: VALU %.3g, SALU %.3g, LDS %.3g, vector loads %.3g" % (c["SQ_INSTS_VALU"], c.get("SQ_INSTS_SALU", 0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_VMEM_RD", 0)))


if __name__ == "__main__":
    main()
