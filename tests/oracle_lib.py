"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

SCENES = {
    "random": 0, "two-spheres": 1, "two-perlin-spheres": 2, "earth": 3, "simple-light": 4,
    "cornell-box": 5, "cornell-smoke": 6, "final-scene": 7, "mega-spheres": 8,
}

COUNTER_NAMES = [
    "rays", "camera_rays", "aabb", "sphere_miss", "sphere_hit", "moving_miss", "moving_hit", "rect_miss", "rect_hit",
    "translate", "rotate", "medium", "sc_lambert", "sc_metal", "sc_dielectric", "sc_light", "sc_iso",
    "tx_solid", "tx_checker", "tx_noise", "tx_image", "bg", "rng_words", "depth_exhausted",
]


class Ray(C.Structure):
    _fields_ = [("o", C.c_double * 3), ("d", C.c_double * 3), ("time", C.c_double), ("tmin", C.c_double), ("tmax", C.c_double),
                ("pixel", C.c_uint32), ("sample", C.c_uint32), ("bounce", C.c_uint32), ("pad", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("hit", C.c_int32), ("obj", C.c_int32), ("sub", C.c_int32), ("side", C.c_int32), ("material", C.c_int32),
                ("front_face", C.c_int32), ("t", C.c_double), ("p", C.c_double * 3), ("n", C.c_double * 3), ("u", C.c_double),
                ("v", C.c_double)]


RAY_DTYPE = np.dtype([("o", "<f8", 3), ("d", "<f8", 3), ("time", "<f8"), ("tmin", "<f8"), ("tmax", "<f8"),
                      ("pixel", "<u4"), ("sample", "<u4"), ("bounce", "<u4"), ("pad", "<u4")])
HIT_DTYPE = np.dtype([("hit", "<i4"), ("obj", "<i4"), ("sub", "<i4"), ("side", "<i4"), ("material", "<i4"), ("front_face", "<i4"),
                      ("t", "<f8"), ("p", "<f8", 3), ("n", "<f8", 3), ("u", "<f8"), ("v", "<f8")])
assert RAY_DTYPE.itemsize == C.sizeof(Ray) and HIT_DTYPE.itemsize == C.sizeof(Hit)


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH) for f in ("oracle_capi.cpp", "rt_oracle.hpp")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 6
        L.orc_scene_camera.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_scene_object_boxes.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_scene_root_box.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_camera_rays.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_first_hit.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_first_hit.restype = C.c_int
        L.orc_first_hit_runner_up.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_first_hit_runner_up.restype = C.c_int
        L.orc_render.restype = C.c_double
        L.orc_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                 C.c_int]
        L.orc_sample_radiance.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_resolve.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
        L.orc_aabb_hit.argtypes = [C.c_void_p] * 4 + [C.c_double, C.c_double]
        L.orc_sphere_uv.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_philox.argtypes = [C.c_void_p] * 3
        L.orc_philox_rounds.argtypes = [C.c_void_p] * 3 + [C.c_int]
        L.orc_stream_words.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_pcg32_words.argtypes = [C.c_uint64, C.c_int, C.c_void_p]
        L.orc_shape_hit.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_reflect.argtypes = [C.c_void_p] * 3
        L.orc_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        L.orc_schlick.restype = C.c_double
        L.orc_schlick.argtypes = [C.c_double, C.c_double]
        L.orc_color_to_rgb.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_divide_into_ranges.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
        L.orc_material_texture_value.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    """A scene built by the oracle's own restatement of scenes.rs."""

    def __init__(self, name, width=0, aspect=0.0, spp=0, depth=0, aux_seed=1, bvh_seed=7, earth=None, mega_grid=0, want_f32=False, mega_view=0):
        self.name = name
        self._earth = None
        ep, ew, eh = None, 0, 0
        if earth is not None:
            self._earth = np.ascontiguousarray(earth, dtype=np.uint8)
            eh, ew = self._earth.shape[:2]
            ep = _p(self._earth)
        self.h = lib().orc_scene_create(SCENES[name], width, float(aspect), spp, depth, aux_seed, bvh_seed, ep, ew, eh, mega_grid, mega_view,
                                        1 if want_f32 else 0)
        if not self.h:
            raise RuntimeError("orc_scene_create failed")
        vals = [C.c_int() for _ in range(6)]
        lib().orc_scene_info(self.h, *[C.byref(v) for v in vals])
        self.width, self.height, self.spp, self.depth, self.n_objects, self.n_materials = [v.value for v in vals]

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_destroy(self.h)
            self.h = None

    def camera(self):
        out = np.zeros(24)
        lib().orc_scene_camera(self.h, _p(out))
        return out

    def object_boxes(self):
        out = np.zeros((self.n_objects, 6))
        lib().orc_scene_object_boxes(self.h, _p(out))
        return out

    def root_box(self):
        out = np.zeros(6)
        lib().orc_scene_root_box(self.h, _p(out))
        return out

    def camera_rays(self, seed, pixel, sample):
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        sample = np.ascontiguousarray(sample, dtype=np.uint32)
        rays = np.zeros(pixel.size, dtype=RAY_DTYPE)
        lib().orc_camera_rays(self.h, seed, pixel.size, _p(pixel), _p(sample), _p(rays))
        return rays

    def first_hit(self, rays, seed=0, precision=1):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.size, dtype=HIT_DTYPE)
        rc = lib().orc_first_hit(self.h, seed, rays.size, _p(rays), _p(hits), precision)
        if rc:
            raise RuntimeError("orc_first_hit failed (f32 scene not built?)")
        return hits

    def runner_up(self, rays, winners, class_of_obj, seed=0):
        """The closest hit with each ray's winner (and the exact copies of its object) left out."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        winners = np.ascontiguousarray(winners, dtype=HIT_DTYPE)
        cls = np.ascontiguousarray(class_of_obj, dtype=np.int32)
        assert cls.size >= self.n_objects
        hits = np.zeros(rays.size, dtype=HIT_DTYPE)
        lib().orc_first_hit_runner_up(self.h, seed, rays.size, _p(rays), _p(winners), _p(cls), _p(hits))
        return hits

    def render(self, s_begin, s_end, seed=0, threads=None, rows=None, precision=1, accum=None):
        """Returns (accum[H,W,3] float64 linear sums, counters dict, seconds)."""
        if threads is None:
            threads = max(1, (os.cpu_count() or 2) - 1)  # render.rs:184
        if accum is None:
            accum = np.zeros((self.height, self.width, 3))
        cnt = np.zeros(lib().orc_counters_len(), dtype=np.uint64)
        r0, r1 = rows if rows is not None else (0, self.height)
        secs = lib().orc_render(self.h, s_begin, s_end, seed, threads, _p(accum), _p(cnt), r0, r1, precision)
        if secs < 0:
            raise RuntimeError("orc_render failed")
        return accum, dict(zip(COUNTER_NAMES, [int(x) for x in cnt])), secs

    def sample_radiance(self, seed, pixel, sample):
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        sample = np.ascontiguousarray(sample, dtype=np.uint32)
        out = np.zeros((pixel.size, 4))  # r, g, b, closest-hit queries of the path
        lib().orc_sample_radiance(self.h, seed, pixel.size, _p(pixel), _p(sample), _p(out))
        return out

    def sample_path(self, seed, pixel, sample, max_vertices=64):
        """Closest-hit queries of one path: rows of (origin[3], direction[3], t, obj, sub, hit, front_face)."""
        out = np.zeros((max_vertices, 11))
        lib().orc_sample_path.restype = C.c_int
        n = lib().orc_sample_path(C.c_void_p(self.h), C.c_uint64(seed), C.c_uint32(int(pixel)), C.c_uint32(int(sample)), C.c_int(max_vertices), _p(out))
        return out[:min(n, max_vertices)]

    def texture_value(self, material, u, v, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        out = np.zeros(3)
        rc = lib().orc_material_texture_value(self.h, material, u, v, _p(p), _p(out))
        assert rc == 0
        return out


MUTATIONS = {"none": 0, "one-sided-light": 1, "divide-by-w-h": 2, "tmin-zero": 3}


class mutation:
    """Context manager: the oracle with one deliberate fault switched on (a check of the tests, not of the oracle):
    one-sided-light = DiffuseLight emits from its front face only (the reference: both, src/material.rs:160-162);
    divide-by-w-h = pixel coordinates divided by W and H (the reference: W - 1 and H - 1, src/render.rs:232-233);
    tmin-zero = closest-hit interval opens at 0 (the reference: 0.001, src/render.rs:158)."""

    def __init__(self, name):
        self.kind = MUTATIONS[name]

    def __enter__(self):
        lib().orc_set_mutation(self.kind)

    def __exit__(self, *exc):
        lib().orc_set_mutation(0)
        return False


def resolve(accum, spp):
    accum = np.ascontiguousarray(accum, dtype=np.float64)
    rgb = np.zeros(accum.shape, dtype=np.uint8)
    lib().orc_resolve(_p(accum), accum.size // 3, spp, _p(rgb))
    return rgb


def aabb_hit(mn, mx, o, d, tmin, tmax):
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (mn, mx, o, d)]
    return bool(lib().orc_aabb_hit(*[_p(x) for x in a], tmin, tmax))


def sphere_uv(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    u, v = C.c_double(), C.c_double()
    lib().orc_sphere_uv(_p(p), C.byref(u), C.byref(v))
    return u.value, v.value


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox(_p(c), _p(k), _p(out))
    return out


def philox_rounds(ctr, key, rounds):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox_rounds(_p(c), _p(k), _p(out), rounds)
    return out


def philox_stream_rounds():
    return lib().orc_philox_stream_rounds()


def stream_words(seed, pixel, sample, bounce, stage, n):
    out = np.zeros(n, dtype=np.uint64)
    lib().orc_stream_words(seed, pixel, sample, bounce, stage, n, _p(out))
    return out


def pcg32_words(seed, n):
    out = np.zeros(n, dtype=np.uint32)
    lib().orc_pcg32_words(seed, n, _p(out))
    return out


def shape_hit(kind, params, o, d, tmin=0.001, tmax=float("inf"), time=0.0):
    prm = np.ascontiguousarray(params, dtype=np.float64)
    ray = np.zeros(1, dtype=RAY_DTYPE)
    ray["o"][0] = o
    ray["d"][0] = d
    ray["time"], ray["tmin"], ray["tmax"] = time, tmin, tmax
    hit = np.zeros(1, dtype=HIT_DTYPE)
    lib().orc_shape_hit(kind, _p(prm), _p(ray), _p(hit))
    return hit[0]


def reflect(v, n):
    v, n = np.ascontiguousarray(v, dtype=np.float64), np.ascontiguousarray(n, dtype=np.float64)
    out = np.zeros(3)
    lib().orc_reflect(_p(v), _p(n), _p(out))
    return out


def refract(uv, n, ratio):
    uv, n = np.ascontiguousarray(uv, dtype=np.float64), np.ascontiguousarray(n, dtype=np.float64)
    out = np.zeros(3)
    lib().orc_refract(_p(uv), _p(n), ratio, _p(out))
    return out


def schlick(cosine, ref_idx):
    return lib().orc_schlick(cosine, ref_idx)


def color_to_rgb(rgb_sum, spp):
    s = np.ascontiguousarray(rgb_sum, dtype=np.float64)
    out = np.zeros(3, dtype=np.uint8)
    lib().orc_color_to_rgb(_p(s), spp, _p(out))
    return out


def divide_into_ranges(rows, ranges):
    out = np.zeros((max(rows, ranges, 1), 2), dtype=np.uint32)
    n = lib().orc_divide_into_ranges(rows, ranges, _p(out), out.shape[0])
    return [tuple(int(x) for x in r) for r in out[:n]]
