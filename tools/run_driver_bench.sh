#!/bin/bash
# The driver's round-end bench command, timed:  tools/run_driver_bench.sh <out-dir under gpurun_out/>
out=$1; mkdir -p "$out"
( time timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err" ) 2> "$out/time.txt" || { tail -5 "$out/bench.err"; exit 1; }
grep real "$out/time.txt"
python3 tools/show_bench_line.py "$out/bench.json"
