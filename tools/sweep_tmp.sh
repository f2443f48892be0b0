timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
for st in 1 2 3; do
echo "== streams $st"
timeout -k 5 100 python tools/prof_render.py --spp 64 --depth 8 --repeat 3 --opt streams=$st | tail -1 || exit 1
timeout -k 5 100 python tools/prof_render.py --spp 256 --depth 8 --repeat 2 --opt streams=$st | tail -1 || exit 1
timeout -k 5 100 python tools/prof_render.py --scene dragon-spec --width 1024 --spp 16 --depth 8 --repeat 3 --opt streams=$st | tail -1 || exit 1
done
