"""Statistics of the reference's committed sample renders (tests/golden/*_stats.json, written by tests/golden/make_png_stats.py)
and the bands within which a render of the committed scene code is expected to fall.

What each image can pin (README.md:24-32 shows them; nothing ties them to the committed source):
  cornell_box.png   deterministic scene, matches the committed constants: channel means and 8x8 block means (tight).
  earth.jpg         the globe matches (texture, camera, sphere); the BACKGROUND does not: the image shows a gradient sky, the
                    committed scene table has Background::Solid(0.70, 0.80, 1.00) (src/scenes.rs:74-77), so the image predates
                    that line.  Pinned: the block means of the blocks inside the globe; the sky is checked against the constant.
  perlin.png        same sky mismatch, and the marble pattern depends on Perlin tables drawn from the unseeded thread_rng
                    (src/perlin.rs:14-30): only the channel means, loosely.
  simple_light.png  Perlin tables again, but the image is dominated by the light's falloff: channel means and block means, loosely.
  random.png        1920x1080 (the committed constants give 400x225), albedos from the unseeded thread_rng (src/scenes.rs:172,185):
                    channel means and the block means of the sky / ground structure, loosely.
Bands: tools/pin_noise.py renders every scene under eight seeds for what the reference draws from the unseeded thread_rng and
records, per block of the 9x16 grid, the spread of the block mean over those seeds (tests/golden/block_spread.json; its printout
is kept as profiles/r3_pin_noise.txt).  A block of the image may differ from a render by FLOOR + 4 x that spread: blocks that no
seed touches (sky, far ground, the big spheres' surroundings in random.png; everything outside the marble in simple_light.png) are
pinned at the floor, 0.012 of full scale (three to four times what those blocks measure against the images: <= 0.003 for
random.png, <= 0.004 for simple_light.png), and blocks that depend on the seeds as loosely as the seeds make them.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FLOOR = 0.012

# scene name -> (fixture, band on the 8-bit channel means or None)
FIXTURES = {
    "earth": ("earth_stats.json", None),                # globe blocks only, see globe_blocks()
    "two-perlin-spheres": ("perlin_stats.json", 9.0),
    "simple-light": ("simple_light_stats.json", 1.5),   # spread over seeds 0.28, image - render 0.13
    "random": ("random_stats.json", 8.0),               # spread over seeds 1.8, image - render up to 2.3
}


def block_band(name):
    with open(os.path.join(HERE, "golden", "block_spread.json")) as f:
        sd = np.array(json.load(f)[name]["block_std_worst_channel"])
    return FLOOR + 4.0 * sd


def load(name):
    with open(os.path.join(HERE, "golden", FIXTURES[name][0])) as f:
        return json.load(f)


def image_stats(rgb):
    """(channel means of the 8-bit image, 9x16 block means of linear radiance) of an RGB8 image whose sides divide by 9 and 16.
    linear = (value / 256)^2 undoes color_to_rgb's gamma (src/color.rs:20-25), as the fixtures did."""
    rgb = np.asarray(rgb, dtype=np.float64)
    h, w = rgb.shape[:2]
    lin = (rgb / 256.0) ** 2
    return rgb.reshape(-1, 3).mean(0), lin.reshape(9, h // 9, 16, w // 16, 3).mean(axis=(1, 3))


def globe_blocks():
    """Blocks of the 9x16 grid that lie inside the globe of the earth scene (radius 2 seen from 13.5 away at 20 degrees:
    the disc covers 0.84 of the frame height around the centre)."""
    rows, cols = np.mgrid[0:9, 0:16]
    cy, cx = (rows + 0.5) * 25 - 112.5, (cols + 0.5) * 25 - 200.0
    # a block is inside when its farthest corner is: half-diagonal of a 25-pixel block = 17.7 pixels
    return np.hypot(cx, cy) + 17.7 < 0.42 * 225


def check(name, rgb):
    """Asserts the bands of FIXTURES[name]; returns the measured deviations for printing."""
    g = load(name)
    _, mean_band = FIXTURES[name]
    m8, blocks = image_stats(rgb)
    ref_blocks = np.array(g["block_means_linear"])
    out = {"mean8": m8, "ref_mean8": np.array(g["mean_8bit"])}
    if mean_band is not None:
        assert np.abs(m8 - out["ref_mean8"]).max() < mean_band, (name, m8, g["mean_8bit"])
    if name == "earth":
        inside = globe_blocks()
        assert inside.sum() >= 12
        d = np.abs(blocks - ref_blocks)[inside]
        out["globe_block_max"] = d.max()
        assert d.max() < 0.03, (name, d.max())  # measured 0.022: a JPEG of a render through this library's own decode of the texture
        # the sky: the committed constant (0.70, 0.80, 1.00), clamped and quantised by color_to_rgb
        sky = blocks[0, 0]
        want = (np.floor(256 * np.minimum(np.sqrt([0.70, 0.80, 1.00]), 0.999)) / 256.0) ** 2
        assert np.abs(sky - want).max() < 0.01, (sky, want)
        # ... which is NOT the reference image's sky (a gradient): recorded so that a change of either side shows up
        out["sky_vs_image"] = np.abs(sky - ref_blocks[0, 0]).max()
        assert 0.01 < out["sky_vs_image"] < 0.08
    elif name == "two-perlin-spheres":
        # sky blocks (no seed changes them): the committed constant, as for earth, and NOT the image's gradient sky
        band = block_band(name)
        sky = band <= FLOOR
        assert sky.sum() >= 16
        want = (np.floor(256 * np.minimum(np.sqrt([0.70, 0.80, 1.00]), 0.999)) / 256.0) ** 2
        assert np.abs(blocks[sky] - want).max() < 0.005, np.abs(blocks[sky] - want).max()
        out["sky_vs_image"] = np.abs(blocks - ref_blocks).max(axis=2)[sky].max()
        assert 0.03 < out["sky_vs_image"] < 0.08
        # the marble: as loosely as the seeds make it
        d = np.abs(blocks - ref_blocks).max(axis=2)
        assert (d[~sky] < band[~sky] + 0.06).all(), (name, (d - band)[~sky].max())
    else:
        band = block_band(name)
        d = np.abs(blocks - ref_blocks).max(axis=2)
        out["block_max"] = d.max()
        out["pinned_blocks"] = int((band <= FLOOR + 0.012).sum())
        out["pinned_block_max"] = d[band <= FLOOR + 0.012].max()
        assert (d < band).all(), (name, np.argwhere(d >= band).tolist(), (d - band).max())
        assert out["pinned_blocks"] >= (30 if name == "random" else 80)
    return out
