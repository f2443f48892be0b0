#!/bin/bash
# Profiles of the bench command itself, one rocprofv3 run each: kernel trace + stats, then the two
# HBM-traffic counter passes (the tool does not split counter sets over passes).  The counter passes run one
# batch at a time (--streams 1), so that every launch they see has the size of the launches bench.py times
# with HIP events and the per-launch averages are comparable.
# usage: tools/profile_bench.sh <out-dir> [bench.py arguments]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
ARGS="$*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python bench.py $ARGS --no-cpu-baseline > "$out/trace.log" 2>&1 || { echo "trace run failed"; tail -5 "$out/trace.log"; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d "$out/$c" -- python bench.py --no-cpu-baseline $ARGS --steps 1 --warmup 0 --streams 1 > "$out/$c.log" 2>&1 || { echo "pass $c failed"; tail -5 "$out/$c.log"; exit 1; }
done
echo "profile_bench: done"
