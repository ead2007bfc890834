#!/bin/bash
# Build container only: pictures of ODD and SMALL sizes, encoded by the reference encoder and decoded by the reference decoder -
# crops (top-left corner) of the reference's own test picture test/data/192x128_kodim15.png:
#   odd191x127  both sizes odd (ceil divisions of every grid, the replicate / zero paddings at the right and bottom edges)
#   odd100x37   coarse grids narrower than 10 columns (the raster-order path of latent.py:240-265), odd height
#   odd18x65    a picture narrower than two wavefront steps, taller than wide
# tests/golden/gen/make_odd_crops.sh   ->   tests/golden/odd*.cool / .npz / .json
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
mkdir -p /tmp/ccgen/in
python3 - <<'PY'
from PIL import Image
im = Image.open('/root/reference/test/data/192x128_kodim15.png')
for (w, h) in [(191, 127), (100, 37), (18, 65)]:
    im.crop((0, 0, w, h)).save(f'/tmp/ccgen/in/k15_{w}x{h}.png')
PY
for s in 191x127 100x37 18x65; do
    "$HERE/encode_presets.sh" odd$s /tmp/ccgen/in/k15_$s.png 1 lop
done
