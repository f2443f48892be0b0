timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
for opts in "--opt shade_variant=0" "--opt shade_variant=1"; do
  echo "== $opts"; timeout -k 5 120 python tools/prof_render.py --spp 64 --depth 8 --repeat 2 --stats --opt batch_paths=8388608 $opts | tail -2 || exit 1
done
