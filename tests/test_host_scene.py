"""Host-side logic without a GPU: the C++ scene builder (scenes.rs / camera.rs / bvh.rs restated into flat arrays)
against the oracle's independent pointer-tree restatement, the C ABI's symbol table, image IO and the CLI."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from scene_pairs import ALL_SCENES, make_pair

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(REPO, "rust-raytracing_amd", "_build")


@pytest.mark.parametrize("name", ALL_SCENES)
def test_host_scene_equals_oracle_scene(rtamd, oracle, earth_rgb, name):
    hs, os_ = make_pair(rtamd, oracle, name, earth=earth_rgb)
    # Camera::new, field by field, bit for bit (src/camera.rs:21-60)
    assert np.array_equal(hs.camera(), os_.camera())
    d = hs.desc
    nodes, objects = hs.nodes(), hs.objects()
    root = nodes[d.root_node]
    assert np.array_equal(np.concatenate([root["bmin"], root["bmax"]]), os_.root_box())
    assert d.n_materials == os_.n_materials
    # world-list objects are those with sub_id == -1; their (0,1) boxes must equal the oracle's bounding_box(0,1)
    top = [i for i in range(d.n_objects) if objects[i]["sub_id"] == -1 and objects[i]["obj_id"] >= 0]
    assert len(top) == os_.n_objects
    assert sorted(int(objects[i]["obj_id"]) for i in top) == list(range(os_.n_objects))


@pytest.mark.parametrize("name", ALL_SCENES)
def test_bvh_invariants(rtamd, name, earth_rgb):
    hs = rtamd.Scene(name, earth=earth_rgb if name in ("earth", "final-scene") else None)
    d = hs.desc
    nodes, objects = hs.nodes(), hs.objects()
    seen_nodes, seen_leaves = set(), []

    def walk(n, depth):
        assert n not in seen_nodes
        seen_nodes.add(n)
        nd = nodes[n]
        for c in (int(nd["left"]), int(nd["right"])):
            if c >= 0:
                child = nodes[c]
                # a node's box contains its children's boxes unless a child box is inverted (negative-radius sphere)
                if (child["bmin"] <= child["bmax"]).all():
                    assert (nd["bmin"] <= child["bmin"]).all() and (nd["bmax"] >= child["bmax"]).all()
                walk(c, depth + 1)
            else:
                seen_leaves.append(~c)
        return depth

    walk(d.root_node, 0)
    top = {i for i in range(d.n_objects) if objects[i]["sub_id"] == -1}
    assert set(seen_leaves) == top                       # every world object is in the tree
    # BVHNode::new makes a one-object node point at the same object twice (src/bvh.rs:38)
    assert all(seen_leaves.count(i) in (1, 2) for i in top)
    # left-then-right median split: tree depth is logarithmic
    assert len(seen_nodes) <= 2 * len(top)


def test_bvh_is_seeded_and_reference_ordered(rtamd):
    a = rtamd.Scene("random", bvh_seed=7).nodes()
    b = rtamd.Scene("random", bvh_seed=7).nodes()
    c = rtamd.Scene("random", bvh_seed=8).nodes()
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    # n == 2: ordered by box.min[axis] (src/bvh.rs:39-46) -> for cornell's 8 objects every leaf pair is ordered on some axis
    cs = rtamd.Scene("cornell-box")
    assert cs.desc.n_nodes == 7 and cs.desc.n_objects == 8


def test_threaded_bvh_build_equals_the_serial_one(rtamd, oracle, monkeypatch):
    """The host builds the reference's tree on several threads (the axis sequence is drawn up front, a node's index follows from
    object counts alone): the node array must be byte for byte what one thread produces, and the root box the oracle's own
    recursive build arrives at.  19 600 objects: above the size at which subtrees are handed to threads."""
    monkeypatch.setenv("OMP_NUM_THREADS", "1")
    one = rtamd.Scene("mega-spheres", mega_grid=70).nodes().tobytes()
    monkeypatch.setenv("OMP_NUM_THREADS", "8")
    hs = rtamd.Scene("mega-spheres", mega_grid=70)
    assert hs.nodes().tobytes() == one
    os_ = oracle.OracleScene("mega-spheres", mega_grid=70)
    root = hs.nodes()[hs.desc.root_node]
    assert np.array_equal(np.concatenate([root["bmin"], root["bmax"]]), os_.root_box())


def test_cornell_geometry_constants(rtamd):
    # src/scenes.rs:341-381
    s = rtamd.Scene("cornell-box")
    sh, ob, mt = s.shapes(), s.objects(), s.materials()
    assert [int(x) for x in sh["tag"]] == [4, 4, 3, 3, 3, 2, 5, 5]
    assert list(sh["p"][2][:5]) == [213, 343, 227, 332, 554]
    assert list(sh["p"][6][:6]) == [0, 0, 0, 165, 330, 165] and list(sh["p"][7][:6]) == [0, 0, 0, 168, 165, 165]
    assert ob["flags"][6] == 3 and list(ob["offset"][6]) == [265, 0, 295] and list(ob["offset"][7]) == [130, 0, 65]
    assert abs(ob["sin_theta"][6] - np.sin(np.radians(15))) < 1e-16 and abs(ob["sin_theta"][7] - np.sin(np.radians(-18))) < 1e-16
    assert [int(x) for x in mt["tag"]] == [0, 0, 0, 3]
    tex = s.textures()
    assert list(tex["color"][3]) == [15, 15, 15]
    sm = rtamd.Scene("cornell-smoke")
    assert int(sm.objects()["flags"][6]) == 7 and sm.objects()["neg_inv_density"][6] == -100.0


def _declared_functions(header):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rth?_[a-z_0-9]+)\s*\(", text)))


def test_c_abi_exports_every_declared_symbol(rtamd):
    """The shared libraries load without a GPU and export every function the headers declare."""
    lib = ctypes.CDLL(os.path.join(BUILD, "librtamd.so"), mode=ctypes.RTLD_GLOBAL)
    host = ctypes.CDLL(os.path.join(BUILD, "librtamd_host.so"))
    decl = _declared_functions("rtamd.h")
    assert set(decl) == set(rtamd.ABI_SYMBOLS), set(decl) ^ set(rtamd.ABI_SYMBOLS)
    for name in decl:
        assert getattr(lib, name) is not None
    hdecl = _declared_functions("rtamd_host.h")
    assert set(hdecl) == set(rtamd.HOST_ABI_SYMBOLS), set(hdecl) ^ set(rtamd.HOST_ABI_SYMBOLS)
    for name in hdecl:
        assert getattr(host, name) is not None
    lib.rt_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.rt_version()


def test_no_cpu_fallback_without_device(rtamd):
    """Without a HIP device the product fails loudly; it never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rtamd.RtError) as e:
        rtamd.Context(0)
    assert e.value.code == 2 and "no HIP device" in str(e.value)
    with pytest.raises(RuntimeError):
        rtamd.render_and_save("cornell-box", "/tmp/should_not_exist.ppm", width=16, spp=1)
    assert not os.path.exists("/tmp/should_not_exist.ppm")


def test_product_does_not_reference_the_oracle():
    """Nothing under rust-raytracing_amd/ or include/ includes, links or loads anything under oracle/."""
    for root in (os.path.join(REPO, "rust-raytracing_amd"), os.path.join(REPO, "include")):
        for dirpath, _, files in os.walk(root):
            if "_build" in dirpath or "__pycache__" in dirpath:
                continue
            for f in files:
                if f.endswith((".h", ".hip", ".cpp", ".py")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    assert "rt_oracle" not in text and "liboracle" not in text and "oracle_lib" not in text, os.path.join(dirpath, f)


def test_image_io_round_trip(rtamd, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    for ext in ("png", "ppm"):
        p = str(tmp_path / ("x." + ext))
        rtamd.save_image(p, img)
        assert np.array_equal(np.asarray(Image.open(p).convert("RGB")), img)
    assert np.array_equal(rtamd.load_ppm(str(tmp_path / "x.ppm")), img)
    # JPEG is lossy: a smooth picture (sizes that are not multiples of the 8x8 block) comes back within a few levels, and one
    # of the reference's own renders, re-encoded, within a small mean error.  Byte parity with the image crate's encoder is
    # unpinned (the crate is not in the reference's tree).
    yy, xx = np.mgrid[0:45, 0:83]
    smooth = np.stack([xx * 3, yy * 5, 255 - xx - yy], -1).astype(np.uint8)
    for ext in ("jpg", "JPEG"):
        p = str(tmp_path / ("s." + ext))
        rtamd.save_image(p, smooth)
        back = np.asarray(Image.open(p).convert("RGB")).astype(int)
        assert back.shape == smooth.shape and np.abs(back - smooth).max() <= 6 and np.abs(back - smooth).mean() < 1.5
    rtamd.save_image(str(tmp_path / "n.jpg"), img)            # noise: every coefficient in use, long zero runs absent
    back = np.asarray(Image.open(str(tmp_path / "n.jpg")).convert("RGB")).astype(int)
    assert back.shape == img.shape and np.abs(back.mean((0, 1)) - img.mean((0, 1))).max() < 3
    flat = np.full((16, 24, 3), 255, np.uint8)                # 0xFF bytes in the stream get stuffed
    rtamd.save_image(str(tmp_path / "f.jpg"), flat)
    assert np.abs(np.asarray(Image.open(str(tmp_path / "f.jpg")).convert("RGB")).astype(int) - 255).max() <= 1
    with pytest.raises(RuntimeError):
        rtamd.save_image(str(tmp_path / "x.bmp"), img)       # unknown extension, like image.save
    with pytest.raises(RuntimeError):
        rtamd.load_ppm(str(tmp_path / "missing.ppm"))


def test_cli_argument_relations():
    """clap relations of src/cli.rs:9-20 and the exit codes of src/main.rs:36-46."""
    exe = os.path.join(BUILD, "rt-render")

    def run(*a):
        return subprocess.run([exe, *a], capture_output=True, text=True)

    r = run("--scene", "cornell-box")
    assert r.returncode == 2 and "--no-gui" in r.stderr
    r = run("--no-gui", "--scene", "cornell-box")
    assert r.returncode == 2 and "--output" in r.stderr
    r = run("--no-gui", "-o", "/tmp/a.ppm", "-s", "not-a-scene")
    assert r.returncode == 2 and "invalid value" in r.stderr
    r = run("--no-gui", "-o", "/tmp/a.ppm", "-s", "cornell-box", "--bogus")
    assert r.returncode == 2
    assert run("--help").returncode == 0
    import torch
    if not torch.cuda.is_available():
        r = run("--no-gui", "-o", "/tmp/a.ppm", "-s", "cornell-box", "--width", "16", "--spp", "1")
        assert r.returncode == 1 and r.stderr.startswith("Error: ")


def test_scene_names(rtamd):
    _, H = rtamd.load_libraries()
    for name, idx in rtamd.SCENES.items():
        assert H.rth_scene_from_name(name.encode()) == idx
        assert H.rth_scene_name(idx).decode() == name
    assert H.rth_scene_from_name(b"Cornell-Box") == -1


def test_missing_rccl_is_reported_not_crashed(tmp_path):
    """A host without RCCL: rt_comm_unique_id / rt_comm_init must return RT_E_UNSUPPORTED with a message, so that a caller can
    fall back to its own exchange.  RTAMD_RCCL_LIB points the loader at a file that does not exist (and at one that is not a
    library); a fresh process, because a library once loaded stays loaded."""
    code = r"""
import ctypes, os, sys
lib = ctypes.CDLL(sys.argv[1])
lib.rt_last_error.restype = ctypes.c_char_p
buf = (ctypes.c_char * 128)()
rc = lib.rt_comm_unique_id(ctypes.byref(buf))
print(rc, lib.rt_last_error().decode())
"""
    junk = tmp_path / "librccl_not_a_library.so"
    junk.write_text("this is not an ELF file\n")
    for target in (str(tmp_path / "no_such_librccl.so"), str(junk)):
        env = dict(os.environ, RTAMD_RCCL_LIB=target)
        r = subprocess.run([os.sys.executable, "-c", code, os.path.join(BUILD, "librtamd.so")], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 0, r.stderr
        rc, msg = r.stdout.strip().split(" ", 1)
        assert int(rc) == 5 and "RCCL is not available" in msg and len(msg) > 30, r.stdout


def test_header_compiles_as_plain_c(tmp_path):
    """include/rtamd.h and include/rtamd_host.h are C headers: a C11 translation unit that includes them and takes the address of
    every declared function compiles with gcc (no C++ in between) and links against the two libraries."""
    names = _declared_functions("rtamd.h") + _declared_functions("rtamd_host.h")
    src = tmp_path / "abi.c"
    src.write_text('#include "rtamd.h"\n#include "rtamd_host.h"\n#include <stdio.h>\n'
                   "int main(void) {\n  const void* f[] = {%s};\n  printf(\"%%d %%s\\n\", (int)(sizeof f / sizeof f[0]), rt_version());\n"
                   "  rt_context* c = 0; return rt_create(&c, 1 << 20) == RT_E_NO_DEVICE ? 0 : 1;\n}\n" % ", ".join("(const void*)%s" % n for n in names))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-Wno-pedantic", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe),
                           "-L", BUILD, "-lrtamd_host", "-lrtamd", "-Wl,-rpath," + BUILD, "-Wl,-rpath-link," + BUILD])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.split()[0] == str(len(names)) and "gfx950" in r.stdout, (r.returncode, r.stdout, r.stderr)


def _build_c_abi_smoke(tmp_path):
    exe = tmp_path / "c_abi_smoke"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(REPO, "include"),
                           os.path.join(REPO, "tests", "c_abi_smoke.c"), "-o", str(exe), "-L", BUILD, "-lrtamd", "-Wl,-rpath," + BUILD,
                           "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe


def test_c_program_links_and_checks_arguments(tmp_path):
    """tests/c_abi_smoke.c - a C11 program that includes include/rtamd.h, fills an rt_scene_desc by hand and links librtamd.so -
    compiles without warnings under -pedantic; its device-free part (argument checks, error string) runs here."""
    exe = _build_c_abi_smoke(tmp_path)
    r = subprocess.run([str(exe), "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c_abi_smoke ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_c_program_renders_on_the_gpu(tmp_path):
    """The same program, all of it: uploads its two-rectangle scene, renders 16x16 in both precisions on GPU 0 and checks status
    codes, statistics and pixels - the C ABI with no Python in between."""
    exe = _build_c_abi_smoke(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "c_abi_smoke ok", (r.returncode, r.stdout, r.stderr)


def test_binding_layouts_match_the_header(rtamd, tmp_path):
    """Every struct a binding has to mirror (the ctypes classes and numpy record types here, the #[repr(C)] structs of a Rust
    host): size and the offset of every field as a C compiler lays out include/rtamd.h, against the Python mirrors - a field
    added to the header without its mirror, or a pad forgotten, shows up here and not as a garbled render."""
    ct = {"rt_render_params": rtamd.RenderParams, "rt_stats": rtamd.Stats, "rth_scene_opts": rtamd.SceneOpts, "rth_render_config": rtamd.RenderConfig}
    dt = {"rt_bvh_node": rtamd.NODE_DTYPE, "rt_object": rtamd.OBJECT_DTYPE, "rt_shape": rtamd.SHAPE_DTYPE, "rt_material": rtamd.MATERIAL_DTYPE,
          "rt_texture": rtamd.TEXTURE_DTYPE, "rt_ray": rtamd.RAY_DTYPE, "rt_hit": rtamd.HIT_DTYPE, "rth_pixel": rtamd.PIXEL_DTYPE}
    sizes = {"rt_scene_desc": ctypes.sizeof(rtamd.SceneDesc), "rt_camera": 24 * 8, "rt_background": 8 + 48, "rt_perlin": 256 * 24 + 3 * 1024,
             "rt_image": 16, "rt_comm_id": 128}
    lines = []
    for cname, cls in ct.items():
        lines.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        lines += ['printf(" %s=%%zu", offsetof(%s, %s));' % (f[0], cname, f[0]) for f in cls._fields_]
        lines.append('printf("\\n");')
    for cname, d in dt.items():
        lines.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        cnames = {"o": "origin", "d": "direction", "n": "normal"}  # the record types abbreviate three names
        lines += ['printf(" %s=%%zu", offsetof(%s, %s));' % (f, cname, cnames.get(f, f)) for f in d.names]
        lines.append('printf("\\n");')
    for cname in sizes:
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
    src = tmp_path / "layout.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "rtamd.h"\n#include "rtamd_host.h"\nint main(void) {\n%s\nreturn 0;\n}\n' % "\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60).stdout.strip().splitlines()
    assert len(out) == len(ct) + len(dt) + len(sizes)
    for line in out:
        cname, size, *fields = line.split()
        if cname in ct:
            cls = ct[cname]
            assert ctypes.sizeof(cls) == int(size), (cname, ctypes.sizeof(cls), size)
            for f in fields:
                fname, off = f.split("=")
                assert getattr(cls, fname).offset == int(off), (cname, fname, getattr(cls, fname).offset, off)
        elif cname in dt:
            assert dt[cname].itemsize == int(size), (cname, dt[cname].itemsize, size)
            for f in fields:
                fname, off = f.split("=")
                assert dt[cname].fields[fname][1] == int(off), (cname, fname)
        else:
            assert sizes[cname] == int(size), (cname, sizes[cname], size)


def test_pixel_stream_is_the_reference_workers_order(rtamd):
    """rth_pixel_stream: every pixel of a frame exactly once, as (x, y from the top, rgb), in the order the reference's workers
    send theirs (src/render.rs:218-241): row blocks of divide_into_ranges (the oracle's restatement of it), each block from its
    last row (largest j, counted from the bottom) to its first, left to right; drained in pieces like check_progress drains."""
    import oracle_lib
    oracle_lib.build()
    rng = np.random.default_rng(6)
    for w, h, workers in ((7, 5, 3), (16, 9, 15), (4, 3, 8), (5, 1, 1)):
        rgb = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        want = []
        for lo, hi in oracle_lib.divide_into_ranges(h, workers):
            for j in range(int(hi) - 1, int(lo) - 1, -1):
                y = h - 1 - j
                want += [(i, y, tuple(rgb[y, i])) for i in range(w)]
        assert len(want) == w * h and len({(x, y) for x, y, _ in want}) == w * h
        whole = rtamd.pixel_stream(rgb, workers)
        assert [(int(p["x"]), int(p["y"]), tuple(p["rgb"])) for p in whole] == want
        pieces, first = [], 0
        while True:
            part = rtamd.pixel_stream(rgb, workers, first=first, max_items=6)
            if len(part) == 0:
                break
            pieces += [(int(p["x"]), int(p["y"]), tuple(p["rgb"])) for p in part]
            first += len(part)
        assert pieces == want
    assert len(rtamd.pixel_stream(np.zeros((2, 2, 3), np.uint8), 0)) == 0  # no workers: nothing (the reference clamps to >= 1)

