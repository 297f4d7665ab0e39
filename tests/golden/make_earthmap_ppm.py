"""Decodes the reference's earthmap.jpg (1920x960 baseline JPEG, opened by ImageTexture::new,
src/texture.rs:98) into the raw RGB8 fixture the host library and the oracle load.

The image crate's jpeg-decoder and PIL's libjpeg may differ by an LSB or two per texel; the fixture is
what BOTH the oracle and the HIP path consume, so parity between them is unaffected.

Run here (needs /root/reference and PIL):  python tests/golden/make_earthmap_ppm.py
"""
import os
import sys

from PIL import Image

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/earthmap.jpg"
repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dst = os.path.join(repo, "rust-raytracing_amd", "assets", "earthmap.ppm")
im = Image.open(src).convert("RGB")
with open(dst, "wb") as f:
    f.write(b"P6\n%d %d\n255\n" % im.size)
    f.write(im.tobytes())
print(dst, im.size)
