#!/bin/bash
# LDS occupancy of one render: all LDS-array cycles vs the extra cycles bank conflicts add.
# usage: tools/pmc_lds.sh <out-dir> [prof_render.py arguments]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$out/lds" -- python tools/prof_render.py "$@" > "$out/lds.log" 2>&1 || { echo "pass failed"; tail -3 "$out/lds.log"; exit 1; }
