#!/bin/bash
# The round's profile set: for each bench workload a kernel trace (+ stats) of the bench command and the two
# HBM-traffic counter passes (tools/profile_bench.sh), then the per-kernel summaries the repository tracks.
# usage (on the GPU box): tools/profile_round.sh <out-dir under gpurun_out/> <tag, e.g. r02>
set -u
out=$1; tag=$2
mkdir -p "$out"
for w in cfg2 dragon teapot_dof; do
  tools/profile_bench.sh "$out/$w" --workload $w --extra none --no-fast-modes --steps 3 --warmup 1 || exit 1
  trace=$(ls "$out/$w"/trace/*/*kernel_trace.csv | head -1)
  stats=$(ls "$out/$w"/trace/*/*kernel_stats.csv | head -1)
  cp "$stats" "$out/${tag}_${w}_bench_kernel_stats.csv"
  n=9; [ "$w" == "cfg2" ] && n=18
  python3 tools/trace_summary.py "$trace" --last-launches $n > "$out/${tag}_${w}_instrumented_step_kernel_stats.csv"
  python3 tools/make_traffic.py "$out/$w" $w "$out/traffic.json" > "$out/${tag}_${w}_traffic.txt"
  grep -h '^{' "$out/$w/trace.log" | tail -1 > "$out/${tag}_${w}_bench_under_trace.json"
done
echo "profile_round: done"
