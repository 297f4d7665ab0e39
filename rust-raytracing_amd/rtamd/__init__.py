"""ctypes binding of the MI355X path-tracing libraries.

  librtamd.so       GPU hot path behind the C ABI of include/rtamd.h
  librtamd_host.so  host side: scenes, camera, BVH build, image IO (include/rtamd_host.h)

There is no CPU fallback anywhere in this package: importing it needs the two
in-tree libraries (built by `make -C rust-raytracing_amd` or
__graft_entry__.build()), and every compute call needs a HIP device.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
BUILD_DIR = os.environ.get("RTAMD_BUILD_DIR") or os.path.join(os.path.dirname(_HERE), "_build")  # override: A/B builds
LIB_PATH = os.path.join(BUILD_DIR, "librtamd.so")
HOST_LIB_PATH = os.path.join(BUILD_DIR, "librtamd_host.so")

F32, F64 = 0, 1
F64_OWN_TREE = F64 | 0x400  # double arithmetic on the tree the library builds for itself (RT_F64_OWN_TREE)
TRACE_AS_RENDERED = 0x100  # rt_trace_batch: records built by the render kernels' own configuration
TRACE_VISITS = 0x200       # ... and u, v = box entries / leaves tested by the walk
PIPELINE_AUTO, PIPELINE_FUSED, PIPELINE_SORTED, PIPELINE_PERSISTENT = 0, 1, 2, 3
VARIANT_FIXED_GRAB, VARIANT_CHUNKED, VARIANT_STEPWISE, VARIANT_DEFERRED_RECORDS, VARIANT_DEVICE_TREE, VARIANT_NARROW_BOXES = 1, 2, 4, 8, 16, 32  # Stats.kernel_variant bits

SCENES = {
    "random": 0, "two-spheres": 1, "two-perlin-spheres": 2, "earth": 3, "simple-light": 4,
    "cornell-box": 5, "cornell-smoke": 6, "final-scene": 7, "mega-spheres": 8,
}

# every symbol include/rtamd.h and include/rtamd_host.h declare
ABI_SYMBOLS = ["rt_create", "rt_destroy", "rt_upload_scene", "rt_render_accum", "rt_resolve", "rt_read_accum", "rt_render", "rt_render_noalloc",
               "rt_progress_begin", "rt_progress_step", "rt_comm_unique_id", "rt_comm_init", "rt_comm_destroy", "rt_reduce_accum",
               "rt_trace_batch", "rt_camera_rays", "rt_sample_radiance", "rt_sample_path", "rt_last_error", "rt_version"]
HOST_ABI_SYMBOLS = ["rth_scene_create", "rth_scene_destroy", "rth_scene_desc", "rth_scene_config", "rth_scene_from_name",
                    "rth_scene_name", "rth_save_image", "rth_load_ppm", "rth_free", "rth_render_and_save", "rth_pixel_stream", "rth_last_error"]


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rtamd error %d: %s" % (code, msg))
        self.code = code


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32),
                ("max_depth", C.c_uint32), ("precision", C.c_int32), ("pipeline", C.c_int32), ("samples_per_pass", C.c_uint32),
                ("seed", C.c_uint64), ("band_rows", C.c_uint32), ("band_stride", C.c_uint32), ("band_offset", C.c_uint32), ("pad", C.c_uint32)]


def band_rows(height, rows, stride, offset):
    """Rows of a frame of `height` rows that belong to caller `offset` of `stride` when bands of `rows` rows are dealt out in turn
    (rt_render_params::band_rows / band_stride / band_offset)."""
    if stride <= 1:
        return np.arange(height)
    y = np.arange(height)
    return y[(y // rows) % stride == offset]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("camera_rays", C.c_uint64), ("seconds", C.c_double), ("bounce_kernel_ms", C.c_double),
                ("bounce_kernel_launches", C.c_uint64), ("gen_kernel_ms", C.c_double), ("passes", C.c_uint64),
                ("lds_bytes", C.c_uint32), ("scene_in_lds", C.c_uint32), ("kernel_variant", C.c_uint32), ("pad", C.c_uint32),
                ("rays_retraced", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class SceneOpts(C.Structure):
    _fields_ = [("scene", C.c_int32), ("width", C.c_int32), ("aspect", C.c_double), ("spp", C.c_int32), ("depth", C.c_int32),
                ("aux_seed", C.c_uint64), ("bvh_seed", C.c_uint64), ("earth_rgb", C.c_void_p), ("earth_w", C.c_int32),
                ("earth_h", C.c_int32), ("mega_grid", C.c_int32), ("mega_view", C.c_int32), ("earth_path", C.c_char_p)]


class RenderConfig(C.Structure):
    _fields_ = [("scene", SceneOpts), ("output", C.c_char_p), ("device", C.c_int32), ("precision", C.c_int32),
                ("pipeline", C.c_int32), ("quiet", C.c_int32), ("seed", C.c_uint64)]


class SceneDesc(C.Structure):
    """Prefix of rt_scene_desc that Python inspects (camera, background, counts, array pointers)."""
    _fields_ = [("camera", C.c_double * 24), ("bg_kind", C.c_int32), ("bg_pad", C.c_int32), ("bg_c0", C.c_double * 3),
                ("bg_c1", C.c_double * 3), ("root_node", C.c_int32), ("n_nodes", C.c_int32), ("n_objects", C.c_int32),
                ("n_shapes", C.c_int32), ("n_materials", C.c_int32), ("n_textures", C.c_int32), ("n_perlins", C.c_int32),
                ("n_images", C.c_int32), ("nodes", C.c_void_p), ("objects", C.c_void_p), ("shapes", C.c_void_p),
                ("materials", C.c_void_p), ("textures", C.c_void_p), ("perlins", C.c_void_p), ("images", C.c_void_p)]


RAY_DTYPE = np.dtype([("o", "<f8", 3), ("d", "<f8", 3), ("time", "<f8"), ("tmin", "<f8"), ("tmax", "<f8"),
                      ("pixel", "<u4"), ("sample", "<u4"), ("bounce", "<u4"), ("pad", "<u4")])
HIT_DTYPE = np.dtype([("hit", "<i4"), ("obj", "<i4"), ("sub", "<i4"), ("side", "<i4"), ("material", "<i4"), ("front_face", "<i4"),
                      ("t", "<f8"), ("p", "<f8", 3), ("n", "<f8", 3), ("u", "<f8"), ("v", "<f8")])
NODE_DTYPE = np.dtype([("bmin", "<f8", 3), ("bmax", "<f8", 3), ("left", "<i4"), ("right", "<i4")])
OBJECT_DTYPE = np.dtype([("flags", "<u4"), ("geom", "<i4"), ("medium_material", "<i4"), ("obj_id", "<i4"), ("sub_id", "<i4"),
                         ("pad", "<i4"), ("offset", "<f8", 3), ("sin_theta", "<f8"), ("cos_theta", "<f8"), ("neg_inv_density", "<f8")])
SHAPE_DTYPE = np.dtype([("tag", "<i4"), ("material", "<i4"), ("p", "<f8", 9)])
MATERIAL_DTYPE = np.dtype([("tag", "<i4"), ("texture", "<i4"), ("albedo", "<f8", 3), ("fuzz", "<f8"), ("ir", "<f8")])
TEXTURE_DTYPE = np.dtype([("tag", "<i4"), ("even", "<i4"), ("odd", "<i4"), ("perlin", "<i4"), ("image", "<i4"), ("pad", "<i4"),
                          ("color", "<f8", 3), ("scale", "<f8")])

_lib = None
_host = None


def load_libraries():
    """Loads both libraries or raises: the product has no fallback when the HIP extension is missing."""
    global _lib, _host
    if _lib is not None:
        return _lib, _host
    for p in (LIB_PATH, HOST_LIB_PATH):
        if not os.path.exists(p):
            raise ImportError("%s is missing: build it with `make -C rust-raytracing_amd` (hipcc, gfx950)" % p)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    H = C.CDLL(HOST_LIB_PATH)
    vp = C.c_void_p
    L.rt_create.argtypes = [C.POINTER(vp), C.c_int]
    L.rt_destroy.argtypes = [vp]
    L.rt_destroy.restype = None
    L.rt_upload_scene.argtypes = [vp, vp]
    L.rt_render_accum.argtypes = [vp, C.POINTER(RenderParams), vp, C.POINTER(Stats)]
    L.rt_resolve.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp]
    L.rt_read_accum.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_int, vp]
    L.rt_render.argtypes = [vp, C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]
    L.rt_render_noalloc.argtypes = [vp, C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]
    L.rt_progress_begin.argtypes = [vp, C.POINTER(RenderParams)]
    L.rt_progress_step.argtypes = [vp, C.c_uint32, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(Stats)]
    L.rt_comm_unique_id.argtypes = [vp]
    L.rt_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.rt_comm_destroy.argtypes = [vp]
    L.rt_reduce_accum.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
    L.rt_trace_batch.argtypes = [vp, vp, C.c_size_t, vp, C.c_int, C.c_uint64]
    L.rt_camera_rays.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, C.c_size_t, C.c_uint64, C.c_int, vp]
    L.rt_sample_radiance.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, C.c_size_t, C.c_uint64, C.c_int, vp]
    L.rt_sample_path.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, C.c_size_t, C.c_uint64, C.c_int, C.c_uint32, vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_version.restype = C.c_char_p
    H.rth_scene_create.restype = vp
    H.rth_scene_create.argtypes = [C.POINTER(SceneOpts)]
    H.rth_scene_destroy.argtypes = [vp]
    H.rth_scene_destroy.restype = None
    H.rth_scene_desc.restype = vp
    H.rth_scene_desc.argtypes = [vp]
    H.rth_scene_config.argtypes = [vp] + [C.POINTER(C.c_uint32)] * 4
    H.rth_scene_config.restype = None
    H.rth_scene_from_name.argtypes = [C.c_char_p]
    H.rth_scene_name.restype = C.c_char_p
    H.rth_scene_name.argtypes = [C.c_int]
    H.rth_save_image.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    H.rth_load_ppm.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    H.rth_free.argtypes = [vp]
    H.rth_free.restype = None
    H.rth_render_and_save.argtypes = [C.POINTER(RenderConfig), C.POINTER(Stats)]
    H.rth_last_error.restype = C.c_char_p
    _lib, _host = L, H
    return L, H


def _check(rc):
    if rc != 0:
        raise RtError(rc, load_libraries()[0].rt_last_error().decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Scene:
    """SceneConfig::get_scene (src/scenes.rs:44-133) through the C++ host library."""

    def __init__(self, name, width=0, aspect=0.0, spp=0, depth=0, aux_seed=1, bvh_seed=7, earth=None, earth_path=None, mega_grid=0, mega_view=0):
        _, H = load_libraries()
        self.name = name
        o = SceneOpts()
        o.scene = SCENES[name] if isinstance(name, str) else int(name)
        o.width, o.aspect, o.spp, o.depth = width, float(aspect), spp, depth
        o.aux_seed, o.bvh_seed, o.mega_grid, o.mega_view = aux_seed, bvh_seed, mega_grid, mega_view
        self._earth = None
        if earth is not None:
            self._earth = np.ascontiguousarray(earth, dtype=np.uint8)
            o.earth_rgb = self._earth.ctypes.data
            o.earth_h, o.earth_w = self._earth.shape[:2]
        self._earth_path = earth_path.encode() if earth_path else None
        o.earth_path = self._earth_path
        self.h = H.rth_scene_create(C.byref(o))
        if not self.h:
            raise RuntimeError("rth_scene_create: " + H.rth_last_error().decode())
        v = [C.c_uint32() for _ in range(4)]
        H.rth_scene_config(self.h, *[C.byref(x) for x in v])
        self.width, self.height, self.spp, self.depth = [x.value for x in v]
        self.desc_ptr = H.rth_scene_desc(self.h)

    def __del__(self):
        if getattr(self, "h", None) and _host is not None:
            _host.rth_scene_destroy(self.h)
            self.h = None

    @property
    def desc(self):
        return SceneDesc.from_address(self.desc_ptr)

    def _array(self, ptr, n, dtype):
        if not ptr or n <= 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_uint8 * (n * dtype.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=n).copy()  # the scene owns the memory; never hand out a view

    def nodes(self):
        d = self.desc
        return self._array(d.nodes, d.n_nodes, NODE_DTYPE)

    def objects(self):
        d = self.desc
        return self._array(d.objects, d.n_objects, OBJECT_DTYPE)

    def shapes(self):
        d = self.desc
        return self._array(d.shapes, d.n_shapes, SHAPE_DTYPE)

    def materials(self):
        d = self.desc
        return self._array(d.materials, d.n_materials, MATERIAL_DTYPE)

    def textures(self):
        d = self.desc
        return self._array(d.textures, d.n_textures, TEXTURE_DTYPE)

    def camera(self):
        return np.array(self.desc.camera[:], dtype=np.float64)


class Context:
    """One GPU.  Mirrors the lifetime of the thread pool in render() (src/render.rs:184)."""

    def __init__(self, device=0):
        L, _ = load_libraries()
        h = C.c_void_p()
        _check(L.rt_create(C.byref(h), device))
        self.h = h
        self.L = L
        self.scene = None

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def upload(self, scene):
        ptr = scene.desc_ptr if hasattr(scene, "desc_ptr") else scene  # a Scene, anything else that owns an rt_scene_desc, or its address
        _check(self.L.rt_upload_scene(self.h, ptr))
        self.scene = scene

    def _params(self, width, height, s_begin, s_end, depth, precision, pipeline, samples_per_pass, seed, bands=None):
        sc = self.scene
        p = RenderParams()
        p.width = width or sc.width
        p.height = height or sc.height
        p.sample_begin, p.sample_end = s_begin, s_end
        p.max_depth = depth if depth is not None else sc.depth
        p.precision, p.pipeline, p.samples_per_pass, p.seed = precision, pipeline, samples_per_pass, seed
        if bands is not None:  # (rows per band, number of callers sharing the frame, this caller's index)
            p.band_rows, p.band_stride, p.band_offset = bands
        return p

    def render(self, s_begin=0, s_end=None, precision=F64, pipeline=PIPELINE_AUTO, seed=0, samples_per_pass=0, depth=None, width=0,
               height=0, want_accum=True, want_rgb=True, bands=None):
        """rt_render: returns (rgb8[H,W,3] or None, accum[H,W,3] float64 or None, stats dict)."""
        if s_end is None:
            s_end = self.scene.spp
        p = self._params(width, height, s_begin, s_end, depth, precision, pipeline, samples_per_pass, seed, bands)
        rgb = np.zeros((p.height, p.width, 3), dtype=np.uint8) if want_rgb else None
        acc = np.zeros((p.height, p.width, 3), dtype=np.float64) if want_accum else None
        st = Stats()
        _check(self.L.rt_render(self.h, C.byref(p), _p(rgb) if want_rgb else None, _p(acc) if want_accum else None, C.byref(st)))
        return rgb, acc, st.as_dict()

    def render_accum(self, accum_dev_ptr, s_begin, s_end, precision=F64, pipeline=PIPELINE_AUTO, seed=0, samples_per_pass=0, depth=None,
                     width=0, height=0, bands=None):
        """rt_render_accum into a caller-owned DEVICE buffer (address as int) or the context's own (None)."""
        p = self._params(width, height, s_begin, s_end, depth, precision, pipeline, samples_per_pass, seed, bands)
        st = Stats()
        _check(self.L.rt_render_accum(self.h, C.byref(p), C.c_void_p(accum_dev_ptr) if accum_dev_ptr else None, C.byref(st)))
        return st.as_dict()

    def render_noalloc(self, accum_dev_ptr, rgb_out, s_begin, s_end, precision=F64, pipeline=PIPELINE_AUTO, seed=0, depth=None, width=0, height=0):
        """rt_render_noalloc: whole frame into the caller's DEVICE sums (address as int) and host RGB8 array; nothing is allocated."""
        p = self._params(width, height, s_begin, s_end, depth, precision, pipeline, 0, seed)
        assert rgb_out.dtype == np.uint8 and rgb_out.size == p.width * p.height * 3 and rgb_out.flags["C_CONTIGUOUS"]
        st = Stats()
        _check(self.L.rt_render_noalloc(self.h, C.byref(p), C.c_void_p(accum_dev_ptr), _p(rgb_out), C.byref(st)))
        return st.as_dict()

    def progress_begin(self, s_begin=0, s_end=None, precision=F64, pipeline=PIPELINE_AUTO, seed=0, depth=None, width=0, height=0):
        """rt_progress_begin: fixes the frame of a progressive render."""
        p = self._params(width, height, s_begin, self.scene.spp if s_end is None else s_end, depth, precision, pipeline, 0, seed)
        _check(self.L.rt_progress_begin(self.h, C.byref(p)))
        self._progress_shape = (p.height, p.width, 3)

    def progress_step(self, max_samples, out=None):
        """rt_progress_step: renders the next slice; returns (rgb8 of all samples so far, samples done, finished, stats)."""
        rgb = out if out is not None else np.zeros(self._progress_shape, dtype=np.uint8)
        done, fin, st = C.c_uint32(), C.c_int(), Stats()
        _check(self.L.rt_progress_step(self.h, max_samples, _p(rgb), C.byref(done), C.byref(fin), C.byref(st)))
        return rgb, done.value, bool(fin.value), st.as_dict()

    def comm_init(self, comm_id, n_ranks, rank):
        """rt_comm_init: joins this context's GPU to the communicator identified by the 128 bytes of `comm_id`."""
        buf = (C.c_char * 128).from_buffer_copy(bytes(comm_id))
        _check(self.L.rt_comm_init(self.h, C.byref(buf), n_ranks, rank))

    def comm_destroy(self):
        _check(self.L.rt_comm_destroy(self.h))

    def reduce_accum(self, accum_dev_ptr, width, height, precision=F64, root=0):
        """rt_reduce_accum: sums the ranks' DEVICE buffers (address as int, or None for the context's own) onto `root`."""
        _check(self.L.rt_reduce_accum(self.h, C.c_void_p(accum_dev_ptr) if accum_dev_ptr else None, width, height, precision, root))

    def resolve(self, accum_dev_ptr, width, height, spp, precision=F64, out=None):
        """rt_resolve; `out` (uint8 [height, width, 3], e.g. a view of pinned memory) is filled in place when given."""
        rgb = out if out is not None else np.zeros((height, width, 3), dtype=np.uint8)
        assert rgb.dtype == np.uint8 and rgb.shape == (height, width, 3) and rgb.flags["C_CONTIGUOUS"]
        _check(self.L.rt_resolve(self.h, C.c_void_p(accum_dev_ptr) if accum_dev_ptr else None, width, height, spp, precision, _p(rgb)))
        return rgb

    def read_accum(self, accum_dev_ptr, width, height, precision=F64):
        out = np.zeros((height, width, 3), dtype=np.float64)
        _check(self.L.rt_read_accum(self.h, C.c_void_p(accum_dev_ptr) if accum_dev_ptr else None, width, height, precision, _p(out)))
        return out

    def trace_batch(self, rays, precision=F64, seed=0, as_rendered=False, visits=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.size, dtype=HIT_DTYPE)
        flags = (TRACE_AS_RENDERED if as_rendered else 0) | (TRACE_VISITS if visits else 0)
        _check(self.L.rt_trace_batch(self.h, _p(rays), rays.size, _p(hits), precision | flags, seed))
        return hits

    def walk_cost(self, precision, n_paths=8192, seed=0, max_vertices=8):
        """Mean number of (box entries, leaves) a closest-hit query of a render tests on the tree the kernels of `precision` walk:
        the rays of `n_paths` random paths of the uploaded scene's frame (their first max_vertices queries) traced once more with
        RT_TRACE_VISITS."""
        rng = np.random.default_rng(seed)
        sc = self.scene
        pixel = rng.integers(0, sc.width * sc.height, size=n_paths).astype(np.uint32)
        sample = rng.integers(0, max(1, sc.spp), size=n_paths).astype(np.uint32)
        _, verts = self.sample_path(pixel, sample, seed=seed, precision=precision, max_vertices=max_vertices)
        v = verts.reshape(-1, 12)
        used = (np.abs(v[:, 3:6]).sum(axis=1) > 0)  # unused records are zero
        rays = np.zeros(int(used.sum()), dtype=RAY_DTYPE)
        rays["o"], rays["d"] = v[used, 0:3], v[used, 3:6]
        rays["time"] = rng.uniform(0, 1, size=rays.size)
        rays["tmin"], rays["tmax"] = 0.001, np.inf
        h = self.trace_batch(rays, precision=precision, seed=seed, visits=True)
        return float(h["u"].mean()), float(h["v"].mean()), int(rays.size)

    def camera_rays(self, pixel, sample, seed=0, precision=F64, width=0, height=0):
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        sample = np.ascontiguousarray(sample, dtype=np.uint32)
        out = np.zeros(pixel.size, dtype=RAY_DTYPE)
        _check(self.L.rt_camera_rays(self.h, width or self.scene.width, height or self.scene.height, _p(pixel), _p(sample), pixel.size,
                                     seed, precision, _p(out)))
        return out

    def sample_radiance(self, pixel, sample, seed=0, precision=F64, depth=None, width=0, height=0):
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        sample = np.ascontiguousarray(sample, dtype=np.uint32)
        out = np.zeros((pixel.size, 4), dtype=np.float64)
        _check(self.L.rt_sample_radiance(self.h, width or self.scene.width, height or self.scene.height,
                                         depth if depth is not None else self.scene.depth, _p(pixel), _p(sample), pixel.size, seed,
                                         precision, _p(out)))
        return out

    def sample_path(self, pixel, sample, seed=0, precision=F64, max_vertices=8, depth=None, width=0, height=0):
        """Returns (radiance[n,4], vertices[n,max_vertices,12])."""
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        sample = np.ascontiguousarray(sample, dtype=np.uint32)
        n = pixel.size
        out = np.zeros(n * (4 + max_vertices * 12), dtype=np.float64)
        _check(self.L.rt_sample_path(self.h, width or self.scene.width, height or self.scene.height,
                                     depth if depth is not None else self.scene.depth, _p(pixel), _p(sample), n, seed, precision,
                                     max_vertices, _p(out)))
        return out[:n * 4].reshape(n, 4), out[n * 4:].reshape(n, max_vertices, 12)


def comm_unique_id():
    """rt_comm_unique_id: the 128 bytes rank 0 hands to the other ranks."""
    L, _ = load_libraries()
    buf = (C.c_char * 128)()
    _check(L.rt_comm_unique_id(C.byref(buf)))
    return bytes(buf.raw)


PIXEL_DTYPE = np.dtype([("x", "<u4"), ("y", "<u4"), ("rgb", "u1", 3), ("pad", "u1")])  # = rth_pixel


def pixel_stream(rgb, workers, first=0, max_items=None):
    """rth_pixel_stream: the (x, y, rgb) items ThreadedRenderer::check_progress would drain for this frame (src/render.rs:39,84-108)."""
    _, H = load_libraries()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    n = w * h if max_items is None else max_items
    out = np.zeros(n, dtype=PIXEL_DTYPE)
    H.rth_pixel_stream.restype = C.c_size_t
    H.rth_pixel_stream.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, C.c_size_t, C.c_void_p]
    got = H.rth_pixel_stream(_p(rgb), w, h, workers, first, n, _p(out))
    return out[:got]


def save_image(path, rgb):
    _, H = load_libraries()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    rc = H.rth_save_image(path.encode(), _p(rgb), rgb.shape[1], rgb.shape[0])
    if rc:
        raise RuntimeError("rth_save_image: " + H.rth_last_error().decode())


def load_ppm(path):
    _, H = load_libraries()
    ptr, w, h = C.c_void_p(), C.c_int32(), C.c_int32()
    if H.rth_load_ppm(path.encode(), C.byref(ptr), C.byref(w), C.byref(h)):
        raise RuntimeError("rth_load_ppm: " + H.rth_last_error().decode())
    n = w.value * h.value * 3
    out = np.frombuffer((C.c_uint8 * n).from_address(ptr.value), dtype=np.uint8).reshape(h.value, w.value, 3).copy()
    H.rth_free(ptr)
    return out


def render_and_save(scene, output, device=0, precision=F64, pipeline=PIPELINE_AUTO, seed=0, quiet=True, **scene_kw):
    """render_and_save (src/render.rs:278-284) through the host library."""
    _, H = load_libraries()
    cfg = RenderConfig()
    cfg.scene.scene = SCENES[scene]
    cfg.scene.width = scene_kw.get("width", 0)
    cfg.scene.aspect = float(scene_kw.get("aspect", 0.0))
    cfg.scene.spp = scene_kw.get("spp", 0)
    cfg.scene.depth = scene_kw.get("depth", 0)
    cfg.scene.aux_seed = scene_kw.get("aux_seed", 1)
    cfg.scene.bvh_seed = scene_kw.get("bvh_seed", 7)
    cfg.scene.mega_grid = scene_kw.get("mega_grid", 0)
    cfg.scene.mega_view = scene_kw.get("mega_view", 0)
    ep = scene_kw.get("earth_path")
    cfg.scene.earth_path = ep.encode() if ep else None
    cfg.output = output.encode()
    cfg.device, cfg.precision, cfg.pipeline, cfg.quiet, cfg.seed = device, precision, pipeline, 1 if quiet else 0, seed
    st = Stats()
    rc = H.rth_render_and_save(C.byref(cfg), C.byref(st))
    if rc:
        raise RuntimeError("rth_render_and_save: " + H.rth_last_error().decode())
    return st.as_dict()
