#!/bin/bash
# PMC passes for one render, one rocprofv3 run per counter set (the tool does not split passes).
# usage: tools/pmc_passes.sh <out-dir> [prof_render.py arguments]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python tools/prof_render.py $ARGS > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$out/$name.log"; return 1; }
}
ARGS="$*"
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD &&
pass sq2 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_ANY &&
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE &&
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE &&
pass l2 TCC_HIT_sum TCC_MISS_sum
