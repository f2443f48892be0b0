for l in 1 4 32; do
echo "== variant 2 leaf_go $l"
timeout -k 5 100 python tools/prof_render.py --spp 32 --depth 8 --repeat 2 --stats --count --opt extend_variant=2 --opt pw_leaf_go=$l --opt streams=1 | tail -3 || exit 1
timeout -k 5 100 python tools/prof_render.py --scene dragon-spec --width 1024 --spp 8 --depth 8 --repeat 2 --stats --count --opt extend_variant=2 --opt pw_leaf_go=$l --opt streams=1 | tail -3 || exit 1
done
