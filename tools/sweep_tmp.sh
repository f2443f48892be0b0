for sc in dragon-spec dragon-trans teapot-trans; do
for opts in "--opt extend_variant=0" "--opt extend_variant=1"; do
  echo "== $sc $opts"; timeout -k 5 200 python tools/prof_render.py --scene $sc --width 1024 --spp 16 --depth 8 --repeat 2 --stats $opts | tail -2 || exit 1
done; done
