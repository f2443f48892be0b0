#!/bin/bash
# Sweep one cpt_set_option key over values with tools/ab_bench.sh (in-tree library):  tools/sweep_opts.sh key v1 v2 ... [-- bench args]
key=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
for v in "${vals[@]}"; do
  echo "== $key=$v"
  tools/ab_bench.sh default -- --opt $key=$v "$@" || exit 1
done
