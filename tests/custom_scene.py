"""Scenes the host library does not build: a named scene's flattened arrays, edited in numpy and handed to rt_upload_scene as an
rt_scene_desc of their own (the C ABI takes plain arrays; this is what a foreign host would do)."""
import ctypes as C

import numpy as np


class EditedScene:
    """Copies of a host-built scene's arrays that tests may modify before upload.  Keeps the base scene alive (Perlin tables and
    image bytes are still the base scene's memory unless replaced)."""

    def __init__(self, rtamd, base):
        self.rtamd = rtamd
        self.base = base
        self.width, self.height, self.spp, self.depth = base.width, base.height, base.spp, base.depth
        self.nodes, self.objects, self.shapes = base.nodes(), base.objects(), base.shapes()
        self.materials, self.textures = base.materials(), base.textures()
        d = base.desc
        self.n_perlins = d.n_perlins
        self.perlins = None  # bytes of rt_perlin records when replaced
        self._desc = None

    def replicate_perlins(self, copies):
        """n_perlins := copies * n_perlins, every table copied; returns the index offset of the last copy."""
        d = self.base.desc
        rec = 256 * 3 * 8 + 3 * 256 * 4
        raw = (C.c_uint8 * (rec * d.n_perlins)).from_address(d.perlins)
        one = np.frombuffer(raw, dtype=np.uint8).copy()
        self.perlins = np.tile(one, copies)
        self.n_perlins = d.n_perlins * copies
        return d.n_perlins * (copies - 1)

    @property
    def desc_ptr(self):
        src = self.base.desc
        d = self.rtamd.SceneDesc()
        C.memmove(C.addressof(d), C.addressof(src), C.sizeof(d))
        for name, arr in (("nodes", self.nodes), ("objects", self.objects), ("shapes", self.shapes), ("materials", self.materials),
                          ("textures", self.textures)):
            arr = np.ascontiguousarray(arr)
            setattr(self, name, arr)
            setattr(d, name, arr.ctypes.data)
            setattr(d, "n_" + name, arr.size)
        if self.perlins is not None:
            d.perlins = self.perlins.ctypes.data
            d.n_perlins = self.n_perlins
        self._desc = d
        return C.addressof(d)
