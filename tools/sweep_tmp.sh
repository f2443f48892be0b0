timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 5 100 python tools/prof_render.py --spp 64 --depth 8 --repeat 2 --stats --count | tail -4 || exit 1
timeout -k 5 100 python tools/prof_render.py --scene dragon-spec --width 1024 --spp 16 --depth 8 --repeat 2 --stats --count | tail -4 || exit 1
timeout -k 5 100 python tools/prof_render.py --spp 256 --depth 8 --repeat 2 | tail -1
