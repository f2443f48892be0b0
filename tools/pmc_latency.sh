#!/bin/bash
# Latency view of one render: average time a vector-memory and an LDS instruction spends in flight (SQ_INST_LEVEL_* /
# SQ_INSTS_*), the L1's average read latency to L2 and its own, and where the address unit stalls.  One rocprofv3 run
# per counter set.   usage: tools/pmc_latency.sh <out-dir> [prof_render.py arguments]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
ARGS="$*"
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python tools/prof_render.py $ARGS > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$out/$name.log"; return 1; }
}
pass lat1 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES &&
pass lat2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum &&
pass lat3 TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum &&
pass lat4 TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE &&
pass lat5 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE
python3 - "$out" <<'PY'
import csv, glob, os, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, c in tot.items():
    if "extend" not in k and "shade" not in k:
        continue
    print(k)
    for name in sorted(c):
        print("   %-40s %.6g" % (name, c[name]))
    if c.get("SQ_INSTS_VMEM_RD"):
        print("   => avg cycles a vector-memory read is in flight: %.0f" % (c["SQ_INST_LEVEL_VMEM"] / c["SQ_INSTS_VMEM_RD"]))
    if c.get("SQ_INSTS_LDS"):
        print("   => avg cycles an LDS instruction is in flight:     %.0f" % (c["SQ_INST_LEVEL_LDS"] / c["SQ_INSTS_LDS"]))
    if c.get("TCP_TCC_READ_REQ_sum"):
        print("   => avg L1->L2 read latency: %.0f cycles; L1 latency per access %.0f" % (c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"], c["TCP_TCP_LATENCY_sum"] / max(c["TCP_TOTAL_ACCESSES_sum"], 1)))
PY
