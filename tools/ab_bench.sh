#!/bin/bash
# A/B of builds of the GPU library: runs bench.py once per library given on the command line ("default" = the in-tree
# build) and prints Mpaths/s and the per-kernel milliseconds of every workload.  Usage (on the GPU box):
#   tools/ab_bench.sh default cuda_path_tracer_amd/ab/libcpt_amd_x.so ... [-- extra bench.py arguments]
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out/ab
for lib in "${libs[@]}"; do
  tag=$(basename "$lib" .so)
  if [ "$lib" == "default" ]; then unset CPT_AMD_LIBRARY; else export CPT_AMD_LIBRARY="$lib"; fi
  timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fast-modes "$@" > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err || { echo "$tag FAILED"; tail -5 gpurun_out/ab/$tag.err; exit 1; }
  python3 - "$tag" gpurun_out/ab/$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
rows = [("head", d)] + list(d.get("workloads", {}).items())
out = []
for name, r in rows:
    k = r["roofline"]["kernel_ms_per_step"]
    out.append("%s %.0f Mp/s (ext %.1f sh %.1f gen %.1f; frac %.3f)" % (name, r["value"], k["ms_extend"], k["ms_shade"], k["ms_generate"], r["roofline"]["frac"]))
print("%-28s %s" % (sys.argv[1], " | ".join(out)), flush=True)
PY
done
