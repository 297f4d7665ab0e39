"""How tightly the reference's committed sample renders can pin a render of the committed scene code: the noise of a render at
the image's own sample count (spread over seeds) against its distance from the image's statistics (tests/golden/*_stats.json).
Prints the figures the bands in tests/golden_stats.py and tests/test_gpu_render.py are set from and writes the per-block spread over
scene seeds to tests/golden/block_spread.json (a render of THIS library under different seeds for what the reference takes from the
unseeded thread_rng - no reference data in it).
usage: python tools/pin_noise.py [cornell|random|perlin|simple-light|earth ...]"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rust-raytracing_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import rtamd  # noqa: E402
import golden_stats  # noqa: E402


def lin_stats(rgb, rows, cols):
    lin = (rgb.astype(np.float64) / 256.0) ** 2
    h, w = rgb.shape[:2]
    return lin.reshape(-1, 3).mean(0), lin.reshape(rows, h // rows, cols, w // cols, 3).mean(axis=(1, 3))


def cornell(ctx):
    g = json.load(open(os.path.join(REPO, "tests", "golden", "cornell_box_png_stats.json")))
    ref_m, ref_b = np.array(g["mean_linear"]), np.array(g["block_means_linear"])
    sc = rtamd.Scene("cornell-box")
    ctx.upload(sc)
    ms, bs = [], []
    for seed in range(16):
        rgb, _, _ = ctx.render(0, 200, precision=rtamd.F64, seed=1000 + seed, want_accum=False)
        m, b = lin_stats(rgb, 8, 8)
        ms.append(m); bs.append(b)
    ms, bs = np.array(ms), np.array(bs)
    print("cornell-box 600x600x200, 16 seeds:")
    print("  channel means: spread over seeds (std, relative) %s; mean of seeds vs image (relative) %s; worst single seed %s"
          % (np.round(ms.std(0) / ref_m, 5), np.round((ms.mean(0) - ref_m) / ref_m, 5), np.round(np.abs(ms - ref_m).max(0) / ref_m, 5)))
    sd = bs.std(0)
    print("  8x8 block means: largest spread over seeds (std) %.5f; mean of seeds vs image: max |dev| %.5f, in units of the block's own std: max %.2f"
          % (sd.max(), np.abs(bs.mean(0) - ref_b).max(), (np.abs(bs.mean(0) - ref_b) / np.maximum(sd, 1e-4)).max()))
    print("  worst single seed vs image: max |dev| %.5f" % np.abs(bs - ref_b).max())


SPREAD = {}


def seeded(ctx, name, n_seeds=8):
    g = golden_stats.load(name)
    ref_b = np.array(g["block_means_linear"])
    width = 1920 if name == "random" else 400
    bs, ms = [], []
    earth = rtamd.load_ppm(os.path.join(REPO, "rust-raytracing_amd", "assets", "earthmap.ppm")) if name == "earth" else None
    for seed in range(n_seeds):
        sc = rtamd.Scene(name, width=width, aspect=16.0 / 9.0, aux_seed=1 + seed, earth=earth)
        ctx.upload(sc)
        rgb, _, _ = ctx.render(0, sc.spp, precision=rtamd.F64, seed=50 + seed, want_accum=False)
        m8, b = golden_stats.image_stats(rgb)
        bs.append(b); ms.append(m8)
    bs, ms = np.array(bs), np.array(ms)
    sd = bs.std(0).max(axis=2)  # per block, worst channel
    SPREAD[name] = {"seeds": n_seeds, "size": [width, width * 9 // 16], "block_std_worst_channel": np.round(sd, 5).tolist(),
                    "mean8_std": np.round(ms.std(0), 3).tolist()}
    dev = np.abs(bs.mean(0) - ref_b).max(axis=2)
    print("%s, %d scene seeds (tables / albedos) at the image's size and sample count:" % (name, n_seeds))
    print("  8-bit channel means: spread %s, mean of seeds vs image %s" % (np.round(ms.std(0), 2), np.round(ms.mean(0) - np.array(g["mean_8bit"]), 2)))
    print("  block spread over seeds (std, worst channel), 9x16:")
    print(np.array2string(sd, precision=3, suppress_small=True, max_line_width=200))
    print("  |mean of seeds - image| per block:")
    print(np.array2string(dev, precision=3, suppress_small=True, max_line_width=200))


def main():
    ctx = rtamd.Context(0)
    for what in (sys.argv[1:] or ["cornell", "random", "two-perlin-spheres", "simple-light", "earth"]):
        if what == "cornell":
            cornell(ctx)
        else:
            seeded(ctx, what)
    ctx.close()
    if SPREAD:  # what the bands of tests/golden_stats.py are built from
        path = os.path.join(REPO, "tests", "golden", "block_spread.json")
        old = json.load(open(path)) if os.path.exists(path) else {}
        old.update(SPREAD)
        with open(path, "w") as f:
            json.dump(old, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
