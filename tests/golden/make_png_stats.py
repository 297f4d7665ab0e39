"""Summarises the reference's committed sample render cornell_box.png (600x600, the scene's default constants,
src/scenes.rs:85-93) into coarse statistics used as a statistical known-answer: global channel means and 8x8 block
means of linear radiance, where linear = (value/256)^2 undoes color_to_rgb's gamma (src/color.rs:20-25).

Run here (needs /root/reference and PIL):  python tests/golden/make_png_stats.py
"""
import json
import os

import numpy as np
from PIL import Image

here = os.path.dirname(os.path.abspath(__file__))
out = {}
im = np.asarray(Image.open("/root/reference/cornell_box.png").convert("RGB")).astype(np.float64)
lin = (im / 256.0) ** 2
out["source"] = "cornell_box.png 600x600"
out["mean_8bit"] = im.reshape(-1, 3).mean(0).tolist()
out["mean_linear"] = lin.reshape(-1, 3).mean(0).tolist()
out["block_means_linear"] = lin.reshape(8, 75, 8, 75, 3).mean(axis=(1, 3)).tolist()
with open(os.path.join(here, "cornell_box_png_stats.json"), "w") as f:
    json.dump(out, f)
# The other sample renders (README.md:24-32).  Their scenes draw Perlin tables / albedos from the unseeded thread_rng, or went
# through a JPEG encoder, so they pin less than the Cornell box does; a 9 x 16 grid of block means (blocks of 25 x 25 pixels at
# 400 x 225, 120 x 120 at 1920 x 1080: the grid does not depend on the resolution) still places every large feature.
for name in ("simple_light.png", "perlin.png", "earth.jpg", "random.png"):
    a = np.asarray(Image.open("/root/reference/" + name).convert("RGB")).astype(np.float64)
    h, w = a.shape[:2]
    lin2 = (a / 256.0) ** 2
    out2 = {"source": name, "size": [w, h], "mean_8bit": a.reshape(-1, 3).mean(0).tolist(),
            "mean_linear": lin2.reshape(-1, 3).mean(0).tolist(),
            "block_means_linear": lin2.reshape(9, h // 9, 16, w // 16, 3).mean(axis=(1, 3)).tolist()}
    with open(os.path.join(here, os.path.splitext(name)[0] + "_stats.json"), "w") as f:
        json.dump(out2, f)
print("written")
