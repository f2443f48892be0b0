for st in 1 2; do
echo "== streams $st"
timeout -k 5 100 python tools/prof_render.py --spp 64 --depth 8 --repeat 3 --opt streams=$st | tail -1 || exit 1
timeout -k 5 100 python tools/prof_render.py --scene cornell-box2-specular --spp 64 --depth 8 --repeat 3 --opt streams=$st | tail -1 || exit 1
timeout -k 5 100 python tools/prof_render.py --scene dragon-spec --width 1024 --spp 16 --depth 8 --repeat 3 --opt streams=$st | tail -1 || exit 1
done
timeout -k 5 100 python tools/prof_render.py --spp 64 --depth 8 --repeat 2 --stats | tail -1
