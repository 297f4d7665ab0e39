"""Exactly coincident surfaces at a coordinate that no float represents.

Which of two coincident surfaces the reference reports hangs on its box test: a node whose box begins exactly where the running
closest hit lies is skipped (`t_max <= t_min`, src/aabb.rs:40), so the surface inside it is never tested (src/bvh.rs:108-119).  The
device keeps tree boxes as floats rounded outward; for a scene like this one it keeps the caller's doubles beside them and tests a
box the float form lets a ray into once more.  The expectation is the reference's walk restated here over the oracle's own
Aabb::hit and Box::hit (a four-object tree; the oracle builds named scenes only)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = 0.1  # the shared face: not a float


def _scene(rtamd):
    from custom_scene import EditedScene
    base = rtamd.Scene("two-spheres", width=64, spp=1)
    ed = EditedScene(rtamd, base)
    boxes = np.array([[0.0, 0, 0, K, 1, 1],      # A
                      [-1.0, 0, 0, -0.5, 1, 1],  # C
                      [K, 0, 0, 0.2, 1, 1],      # B: shares the face x = K with A
                      [0.3, 0, 0, 0.4, 1, 1]])   # D
    ed.shapes = np.zeros(4, dtype=rtamd.SHAPE_DTYPE)
    ed.shapes["tag"] = 5
    ed.shapes["material"] = [0, 0, 1, 1]
    ed.shapes["p"][:, :6] = boxes
    ed.objects = np.zeros(4, dtype=rtamd.OBJECT_DTYPE)
    ed.objects["geom"] = np.arange(4)
    ed.objects["medium_material"] = -1
    ed.objects["obj_id"] = np.arange(4)
    ed.objects["sub_id"] = -1
    ed.objects["cos_theta"] = 1.0
    ed.materials = np.zeros(2, dtype=rtamd.MATERIAL_DTYPE)
    ed.materials["texture"] = [0, 1]
    ed.nodes = np.zeros(3, dtype=rtamd.NODE_DTYPE)
    tree = [(1, 2, [0, 1, 2, 3]), (-1, -2, [0, 1]), (-3, -4, [2, 3])]  # root = (A, C), (B, D)
    for n, (left, right, members) in enumerate(tree):
        ed.nodes["left"][n], ed.nodes["right"][n] = left, right
        ed.nodes["bmin"][n] = boxes[members, :3].min(axis=0)
        ed.nodes["bmax"][n] = boxes[members, 3:].max(axis=0)
    return ed, boxes, tree


def _reference_walk(oracle, boxes, nodes, tree, o, d):
    """BVHNode::hit over the tree above: (object, t) of the record it leaves, or (-1, inf)."""
    rec = [-1, np.inf]

    def visit(child, t_min, t_max):
        if child < 0:
            obj = -child - 1
            h = oracle.shape_hit(4, boxes[obj], o, d, tmin=t_min, tmax=t_max)
            if h["hit"]:
                rec[0], rec[1] = obj, float(h["t"])
            return bool(h["hit"])
        if not oracle.aabb_hit(nodes["bmin"][child], nodes["bmax"][child], o, d, t_min, t_max):
            return False
        left, right, _ = tree[child]
        hit_left = visit(left, t_min, t_max)
        hit_right = visit(right, t_min, rec[1] if hit_left else t_max)
        return hit_left or hit_right

    visit(0, 0.001, np.inf)
    return rec


def test_shared_face_at_a_non_float_coordinate(gpu, rtamd, oracle, monkeypatch):
    ed, boxes, tree = _scene(rtamd)
    gpu.upload(ed)
    rng = np.random.default_rng(12)
    n = 3000
    rays = np.zeros(n, dtype=rtamd.RAY_DTYPE)
    inside_a = np.arange(n) < 2 * n // 3  # from inside A towards the shared face; the rest from inside B towards it
    rays["o"][:, 0] = np.where(inside_a, rng.uniform(0.005, 0.095, n), rng.uniform(0.105, 0.195, n))
    rays["o"][:, 1:] = rng.uniform(0.3, 0.7, (n, 2))
    rays["d"][:, 0] = np.where(inside_a, 1.0, -1.0) * rng.uniform(0.5, 1.5, n)
    rays["d"][:, 1:] = rng.uniform(-0.2, 0.2, (n, 2))
    rays["tmin"], rays["tmax"] = 0.001, np.inf
    want = np.array([_reference_walk(oracle, boxes, ed.nodes, tree, rays["o"][k], rays["d"][k]) for k in range(n)])
    assert set(np.unique(want[:, 0])) == {0.0, 2.0}  # some rays leave A's record, some B's: the rounding of 1/d decides
    from_a = want[inside_a, 0]
    assert 0.02 < (from_a == 0).mean() < 0.98, (from_a == 0).mean()
    for as_rendered in (False, True):
        got = gpu.trace_batch(rays, precision=rtamd.F64, as_rendered=as_rendered)
        assert (got["hit"] == 1).all()
        assert (got["obj"] == want[:, 0]).all(), (int((got["obj"] != want[:, 0]).sum()), n)
        assert (got["t"] == want[:, 1]).all()
    # what the doubles are there for: with the float boxes alone B's node lets every such ray in, and B, tested last, wins
    monkeypatch.setenv("RTAMD_NO_EXACT_BOXES", "1")
    gpu.upload(ed)
    monkeypatch.delenv("RTAMD_NO_EXACT_BOXES")
    got = gpu.trace_batch(rays, precision=rtamd.F64)
    assert (got["obj"][inside_a] == 2).all() and (got["obj"] != want[:, 0]).any()
    gpu.upload(ed)  # (the context's scene for whoever comes next is restored by their own upload)


def test_path_kernels_agree_on_the_scene(gpu, rtamd):
    """The same scene rendered by the persistent pipeline (whose per-lane walk steps through the list entry by entry and has a box
    test of its own) and by the queue pipeline (one walk per ray): identical ray counts and sums in double precision, with the
    caller's boxes in use in both."""
    ed, _, _ = _scene(rtamd)
    gpu.upload(ed)
    _, a, sa = gpu.render(0, 16, precision=rtamd.F64, seed=3, pipeline=rtamd.PIPELINE_PERSISTENT)
    _, b, sb = gpu.render(0, 16, precision=rtamd.F64, seed=3, pipeline=rtamd.PIPELINE_FUSED)
    assert sa["rays"] == sb["rays"] and sa["rays"] > sa["camera_rays"]  # some paths bounce off the boxes
    assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())
