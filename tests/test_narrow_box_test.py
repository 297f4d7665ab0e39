"""The double-precision four-wide walk tests child boxes in single precision (wide_keys, csrc/device_trace.h): the test may enter a
box the double-precision test would pass by, never the other way round.  Here both tests are restated in numpy - float32 operations
round like the device's, the scaled corrections are exact in float64 before the final rounding, as a fused multiply-add is - and
run on rays built to graze boxes: aimed at points on faces, edges and corners of the box and then moved by a few units in the last
place, from origins up to a million box sizes away, with the closest hit so far set just behind, at and just in front of the box."""
import numpy as np

U22 = np.float32(2.0 ** -22)


def _f64_test(lo, hi, o, inv, t_min, t_max):
    t0 = (lo - o) * inv
    t1 = (hi - o) * inv
    tn = np.maximum(np.minimum(t0, t1).max(axis=1), t_min)
    tf = np.minimum(np.maximum(t0, t1).min(axis=1), t_max)
    return tn < tf


def _f32_test(lo, hi, o, inv, t_min, t_max):
    f = np.float32
    of, iv = o.astype(f), inv.astype(f)
    e = np.abs(of * iv) * U22
    t0 = (lo.astype(f) - of) * iv
    t1 = (hi.astype(f) - of) * iv
    near = np.minimum(t0, t1) - e
    far = np.maximum(t0, t1) + e
    with np.errstate(over="ignore"):
        tn = np.maximum(near.max(axis=1), f(t_min))
        tf = np.minimum(far.min(axis=1), t_max.astype(f))
    tn_low = (tn.astype(np.float64) - 2.0 ** -22 * np.abs(tn.astype(np.float64))).astype(f)
    tf_high = (tf.astype(np.float64) + 2.0 ** -22 * np.abs(tf.astype(np.float64))).astype(f)
    return tn_low < tf_high


def _nudge(x, ulps):
    return (x.view(np.int64) + ulps).view(np.float64)


def test_single_precision_box_test_never_rejects_what_double_precision_enters():
    rng = np.random.default_rng(17)
    n = 400_000
    entered = rejected_both = extra = 0
    for scale_o in (1.0, 1e2, 1e4, 1e6):
        centre = rng.uniform(-1000, 1000, (n, 3))
        half = 10.0 ** rng.uniform(-3, 2, (n, 3))
        lo = (centre - half).astype(np.float32).astype(np.float64)  # node boxes are floats
        hi = (centre + half).astype(np.float32).astype(np.float64)
        # a point of the box surface: every coordinate at lo, at hi, or somewhere between (faces, edges and corners alike)
        pick = rng.integers(0, 3, (n, 3))
        inside = lo + (hi - lo) * rng.uniform(0, 1, (n, 3))
        target = np.where(pick == 0, lo, np.where(pick == 1, hi, inside))
        # every other ray is aimed beside the box, from a hair's breadth to three box sizes away
        beside = (np.arange(n) % 2 == 1)[:, None]
        target = target + beside * rng.choice([-1.0, 1.0], (n, 3)) * (hi - lo) * 10.0 ** rng.uniform(-12, 0.5, (n, 1))
        o = target + rng.normal(size=(n, 3)) * (hi - lo).max(axis=1, keepdims=True) * scale_o
        d = _nudge(target - o, rng.integers(-4, 5, (n, 3)))
        d[d == 0.0] = 1e-300  # (the device sends rays with a zero component through the reference's form of the test)
        inv = 1.0 / d
        ordinary = (np.abs(inv) <= 2.0 ** 40).all(axis=1) & (np.abs(inv) >= 2.0 ** -40).all(axis=1) & (np.abs(o) <= 2.0 ** 40).all(axis=1)
        dist = np.abs((target - o) * inv).max(axis=1)  # parametric distance of the target point, about 1
        t_max = np.where(rng.integers(0, 4, n) == 0, np.inf, _nudge(dist, rng.integers(-3, 4, n)))
        a = _f64_test(lo, hi, o, inv, 0.001, t_max)[ordinary]
        b = _f32_test(lo, hi, o, inv, 0.001, t_max)[ordinary]
        assert not (a & ~b).any(), (scale_o, int((a & ~b).sum()))
        entered += int(a.sum())
        rejected_both += int((~a & ~b).sum())
        extra += int((~a & b).sum())
    # the rays do graze: both outcomes occur in numbers, and the single-precision test rejects most of what double precision rejects
    assert entered > 100_000 and rejected_both > 100_000, (entered, rejected_both, extra)
    print("\nentered by both %d, rejected by both %d, entered by the single-precision test only %d" % (entered, rejected_both, extra))


def test_single_precision_box_test_still_prunes():
    """Ordinary rays against ordinary boxes: what the single-precision test enters beyond the double-precision one is a sliver."""
    rng = np.random.default_rng(5)
    n = 1_000_000
    centre = rng.uniform(-500, 500, (n, 3))
    half = 10.0 ** rng.uniform(-1, 2, (n, 3))
    lo = (centre - half).astype(np.float32).astype(np.float64)
    hi = (centre + half).astype(np.float32).astype(np.float64)
    o = rng.uniform(-600, 600, (n, 3))
    # aimed at a point within three box sizes of the centre: a third of the rays or so pass through
    d = centre + half * rng.uniform(-3, 3, (n, 3)) - o
    inv = 1.0 / d
    t_max = np.where(rng.integers(0, 2, n) == 0, np.inf, rng.uniform(0, 2, n))
    a = _f64_test(lo, hi, o, inv, 0.001, t_max)
    b = _f32_test(lo, hi, o, inv, 0.001, t_max)
    assert not (a & ~b).any()
    assert a.sum() > 100_000 and (~a).sum() > 100_000
    assert (~a & b).sum() <= 1e-4 * (~a).sum(), ((~a & b).sum(), (~a).sum())


def test_unused_slots_are_never_entered():
    """An unused child slot carries a box at +infinity: the comparison must fail whatever the ray's direction signs are."""
    f = np.float32
    inf = np.array([[np.inf] * 3] * 8)
    signs = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
    o = np.full((8, 3), 3.0)
    with np.errstate(invalid="ignore"):
        got = _f32_test(inf, inf, o, signs * 0.7, 0.001, np.full(8, np.inf))
    assert not got.any()
