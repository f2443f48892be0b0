timeout -k 10 400 python -u -m pytest tests/ -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5 || exit 1
for sc in cornell-box2-transmissive; do
  echo "== $sc"; timeout -k 5 200 python tools/prof_render.py --scene $sc --width 512 --spp 64 --depth 8 --repeat 2 --stats | tail -2 || exit 1
done
echo "== dragon"; timeout -k 5 200 python tools/prof_render.py --scene dragon-spec --width 1024 --spp 16 --depth 8 --repeat 2 --stats | tail -2
