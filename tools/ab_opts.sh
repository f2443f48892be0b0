#!/bin/bash
# A/B of (library, options) pairs: tools/ab_opts.sh "lib|opt1=v,opt2=v" ... [-- bench args];  lib "default" = in-tree build
pairs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do pairs+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out/ab
for pr in "${pairs[@]}"; do
  lib=${pr%%|*}; opts=${pr#*|}; [ "$opts" == "$pr" ] && opts=""
  tag=$(basename "$lib" .so)_$(echo "$opts" | tr ',=' '__')
  if [ "$lib" == "default" ]; then unset CPT_AMD_LIBRARY; else export CPT_AMD_LIBRARY="$lib"; fi
  extra=(); IFS=',' read -ra kv <<< "$opts"; for o in "${kv[@]}"; do [ -n "$o" ] && extra+=(--opt "$o"); done
  timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fast-modes "${extra[@]}" "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err || { echo "$tag FAILED"; tail -5 gpurun_out/ab/$tag.err; exit 1; }
  python3 - "$tag" gpurun_out/ab/$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
rows = [("head", d)] + list(d.get("workloads", {}).items())
out = []
for name, r in rows:
    k = r["roofline"]["kernel_ms_per_step"]
    out.append("%s %.0f Mp/s (ext %.1f sh %.1f gen %.1f; frac %.3f)" % (name, r["value"], k["ms_extend"], k["ms_shade"], k["ms_generate"], r["roofline"]["frac"]))
print("%-34s %s" % (sys.argv[1], " | ".join(out)), flush=True)
PY
done
