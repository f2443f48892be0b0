for r in 4 12 24; do for l in 8 20 32 48; do
  echo "== refill $r leaf_go $l"; timeout -k 5 100 python tools/prof_render.py --spp 64 --depth 8 --repeat 2 --stats --opt pw_refill=$r --opt pw_leaf_go=$l | tail -1 || exit 1
done; done
