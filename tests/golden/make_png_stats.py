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
for name in ("simple_light.png", "perlin.png", "earth.jpg", "random.png"):
    a = np.asarray(Image.open("/root/reference/" + name).convert("RGB")).astype(np.float64)
    out2 = {"source": name, "size": [a.shape[1], a.shape[0]], "mean_8bit": a.reshape(-1, 3).mean(0).tolist(),
            "mean_linear": ((a / 256.0) ** 2).reshape(-1, 3).mean(0).tolist()}
    with open(os.path.join(here, os.path.splitext(name)[0] + "_stats.json"), "w") as f:
        json.dump(out2, f)
print("written")
