// Closest hit for the small-scene presets (WALK_UNIFORM) and the single-precision per-lane walk: the walk only keeps the winning
// candidate (t, object, side) and the HitRecord is built once, after the walk, from those three values.
//
// The reference fills a complete HitRecord on every accepted candidate (src/sphere.rs:64-70, src/aarect.rs:99-105,
// src/hittable.rs:229-239,146-148).  Every field of the final record is a function of the ray, the winning primitive
// and its t alone, evaluated here with the same expressions in the same order, so the result is the same bit for bit;
// what is saved is rewriting a dozen registers per accepted candidate in every wave.  The one thing a deferred record
// cannot reproduce is a stale (u, v) left behind by an earlier candidate when the winner is a shape that does not write
// them (moving sphere, medium): presets containing those shapes together with (u, v)-reading textures use the eager walk.
#pragma once
#include <type_traits>

#include "device_trace.h"

#ifdef RT_STAMPS
#define RT_STAMP_IN(stream, k)                                             \
  do {                                                                     \
    if ((stream).stamp_acc) {                                              \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          \
      unsigned long long now_ = __builtin_amdgcn_s_memtime();              \
      (stream).stamp_acc[k] += now_ - *(stream).stamp_t;                   \
      *(stream).stamp_t = now_;                                            \
    }                                                                      \
  } while (0)
#else
#define RT_STAMP_IN(stream, k) do {} while (0)
#endif

namespace rtd {

constexpr int SIDE_MEDIUM = 6;  // candidate produced by a ConstantMedium (sides -1..5 are surfaces)

// Per-ray reciprocal directions.  f64: 1/d for the slab test and true division for the plane distance, as the reference
// writes them.  f32: one hardware reciprocal per component (1 ulp) serves both.
template <class R> struct RayInv;
template <> struct RayInv<double> {
  V3<double> inv, d;
  RT_DEV explicit RayInv(V3<double> dir) : inv(1.0 / dir.x, 1.0 / dir.y, 1.0 / dir.z), d(dir) {}
  // The quotient num / d from the exactly rounded reciprocal the walk already holds (`inv` is 1/d as IEEE division gives it, or
  // the refined hardware estimate, which is the same number): q = num * inv, one correction step q + (num - d q) inv, and the
  // division sequence's own fix-up instruction for zero / infinite / NaN operands.  Bit-identical to num / d on 2^22 random pairs
  // and the special operands (tools/div_probe.hip) at a third of the instructions; a box is six such quotients.
  template <int AXIS> RT_DEV double quot(double num) const {
    const double dk = AXIS == 0 ? d.x : AXIS == 1 ? d.y : d.z, rk = AXIS == 0 ? inv.x : AXIS == 1 ? inv.y : inv.z;
    const double q = num * rk;
    return __builtin_amdgcn_div_fixup(__builtin_fma(__builtin_fma(-q, dk, num), rk, q), dk, num);
  }
};
template <> struct RayInv<float> {
  V3<float> inv;
  RT_DEV explicit RayInv(V3<float> dir) : inv(__builtin_amdgcn_rcpf(dir.x), __builtin_amdgcn_rcpf(dir.y), __builtin_amdgcn_rcpf(dir.z)) {}
  template <int AXIS> RT_DEV float quot(float num) const { return num * (AXIS == 0 ? inv.x : AXIS == 1 ? inv.y : inv.z); }
};

// src/aarect.rs:88-97: distance to the plane, range test, in-plane bounds (closed).  Written without early exits: the six
// comparisons are the reference's own (negated, so a NaN passes them as it does there) and their conjunction decides.
// KEEP: t_out is left alone on a miss (callers that pass their running t_max as t_out); otherwise it receives the plane distance
// whatever the outcome and the caller selects once.
template <class R, int AXIS, class RI, bool KEEP = true>
RT_DEV bool rect_t(R a0, R a1, R b0, R b1, R k, const Ray<R>& ray, const RI& ri, R t_min, R t_max, R& t_out) {
  constexpr int A = (AXIS == 0) ? 1 : 0;
  constexpr int B = (AXIS == 2) ? 1 : 2;
  const R o[3] = {ray.o.x, ray.o.y, ray.o.z};
  const R d[3] = {ray.d.x, ray.d.y, ray.d.z};
  const R t = ri.template quot<AXIS>(k - o[AXIS]);
  const R a = o[A] + t * d[A];
  const R b = o[B] + t * d[B];
  const bool ok = !(t < t_min) & !(t > t_max) & !(a < a0) & !(a > a1) & !(b < b0) & !(b > b1);
  if constexpr (KEEP) { if (ok) t_out = t; } else t_out = t;
  return ok;
}

template <class R> struct RayRcp;
template <class R> RT_DEV bool box_slab_t(const DShape<R>& s, const Ray<R>& ray, const RayRcp<R>& ri, R t_min, R t_max, int self_side, R& t_out, int& side_out);

// hit() of one shape reduced to its t (and the side of a box).  `self_side` as in shape_hit.
template <class R, class C>
RT_DEV bool shape_t(const DShape<R>& s, const Ray<R>& ray, const RayInv<R>& ri, R t_min, R t_max, int self_side, R& t_out, int& side_out) {
  const bool self = Guard<R>::on && self_side != -2;
  bool hit = false;
  switch (s.tag) {
    case SH_SPHERE: if constexpr (C::has(F_SPHERE)) {
      R root;
      if (sphere_roots(V3<R>(s.p[0], s.p[1], s.p[2]), ray, s.p[3], t_min, t_max, root, self)) { t_out = root; side_out = -1; hit = true; }
    } break;
    case SH_MOVING: if constexpr (C::has(F_MOVING)) {
      R root;
      if (sphere_roots(moving_center(s, ray.time), ray, s.p[6], t_min, t_max, root, self)) { t_out = root; side_out = -1; hit = true; }
    } break;
    case SH_RECT_XY: if constexpr (C::has(F_RECT)) { if (!self && rect_t<R, 2>(s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], ray, ri, t_min, t_max, t_out)) { side_out = -1; hit = true; } } break;
    case SH_RECT_XZ: if constexpr (C::has(F_RECT)) { if (!self && rect_t<R, 1>(s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], ray, ri, t_min, t_max, t_out)) { side_out = -1; hit = true; } } break;
    case SH_RECT_YZ: if constexpr (C::has(F_RECT)) { if (!self && rect_t<R, 0>(s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], ray, ri, t_min, t_max, t_out)) { side_out = -1; hit = true; } } break;
    case SH_BOX: if constexpr (C::has(F_BOX) && (sizeof(R) == sizeof(float) || (C::walk == WALK_LANE_WIDE && !C::has(F_EXACT)))) {
      // single precision, and double precision where the walk is compared with the reference by tolerance: the six sides as
      // three slabs, with the reference's side numbering and tie order - a third of the instructions of six rectangle tests, and in
      // double precision no division
      hit = box_slab_t<R>(s, ray, RayRcp<R>(ri.inv, 0), t_min, t_max, self ? self_side : -2, t_out, side_out);
    } else if constexpr (C::has(F_BOX)) {  // six sides in the order of src/box_struct.rs:22-71; a later side wins a tie
      R lox = s.p[0], loy = s.p[1], loz = s.p[2], hix = s.p[3], hiy = s.p[4], hiz = s.p[5];
      R closest = t_max;
      if (!(self && self_side == 0) && rect_t<R, 2>(lox, hix, loy, hiy, hiz, ray, ri, t_min, closest, closest)) { side_out = 0; hit = true; }
      if (!(self && self_side == 1) && rect_t<R, 2>(lox, hix, loy, hiy, loz, ray, ri, t_min, closest, closest)) { side_out = 1; hit = true; }
      if (!(self && self_side == 2) && rect_t<R, 1>(lox, hix, loz, hiz, hiy, ray, ri, t_min, closest, closest)) { side_out = 2; hit = true; }
      if (!(self && self_side == 3) && rect_t<R, 1>(lox, hix, loz, hiz, loy, ray, ri, t_min, closest, closest)) { side_out = 3; hit = true; }
      if (!(self && self_side == 4) && rect_t<R, 0>(loy, hiy, loz, hiz, hix, ray, ri, t_min, closest, closest)) { side_out = 4; hit = true; }
      if (!(self && self_side == 5) && rect_t<R, 0>(loy, hiy, loz, hiz, lox, ray, ri, t_min, closest, closest)) { side_out = 5; hit = true; }
      if (hit) t_out = closest;
    } break;
    default: break;
  }
  return hit;
}

// Ray in the space of the geometry below [Translate(] [RotateY(] (src/hittable.rs:141,214-223).
template <class R> RT_DEV void to_local(const DObject<R>& ob, const Ray<R>& ray, Ray<R>& moved, Ray<R>& local) {
  moved = ray;
  moved.o = ray.o - V3<R>(ob.offset[0], ob.offset[1], ob.offset[2]);  // the offset of an object without Translate is stored as zero: x - 0 = x
  local = moved;
  if (ob.flags & OBJ_ROTATE_Y) {
    local.o.x = ob.cos_theta * moved.o.x - ob.sin_theta * moved.o.z;
    local.o.z = ob.sin_theta * moved.o.x + ob.cos_theta * moved.o.z;
    local.d.x = ob.cos_theta * moved.d.x - ob.sin_theta * moved.d.z;
    local.d.z = ob.sin_theta * moved.d.x + ob.cos_theta * moved.d.z;
    local.time = (R)0;
  }
}

template <class R, class C>
RT_DEV bool boundary_t(const DObject<R>& ob, const DShape<R>& s, const Ray<R>& ray, const RayInv<R>& ri, R t_min, R t_max, int self_side, R& t_out,
                       int& side_out) {
  if constexpr (C::has(F_XFORM)) {
    if (ob.flags & (OBJ_TRANSLATE | OBJ_ROTATE_Y)) {
      Ray<R> moved, local;
      to_local(ob, ray, moved, local);
      if (ob.flags & OBJ_ROTATE_Y) {
        const RayInv<R> rl(local.d);
        return shape_t<R, C>(s, local, rl, t_min, t_max, self_side, t_out, side_out);
      }
      return shape_t<R, C>(s, local, ri, t_min, t_max, self_side, t_out, side_out);  // a translation keeps the direction
    }
  }
  return shape_t<R, C>(s, ray, ri, t_min, t_max, self_side, t_out, side_out);
}

// One leaf of the tree (object index uniform across the wave).
template <class R, class C>
RT_DEV bool leaf_t(const SceneView<R, C>& sv, uint32_t leaf_word, const Ray<R>& ray, const RayInv<R>& ri, R t_min, R t_max, Stream& media, Origin org,
                   R& t_out, int& side_out, uint32_t visit = 0u) {
  const int obj = (int)(leaf_word & LEAF_INDEX_MASK);
  const DShape<R> s = sv.shape(obj);
  const int self_side = org.leaf() == obj ? org.side() : -2;
  if constexpr (C::has(F_XFORM) || C::has(F_MEDIUM)) {
    if (!(leaf_word & LEAF_SIMPLE)) {
      const DObject<R> ob = sv.object(obj);
      if constexpr (C::has(F_MEDIUM)) {
        if (ob.flags & OBJ_MEDIUM) {  // src/constant_medium.rs:41-111
          const R inf = Lim<R>::inf();
          R t1 = (R)0, t2 = (R)0;
          int unused;
          const bool ball = s.tag == SH_SPHERE || s.tag == SH_MOVING;
          if (ball) {  // both crossings from one evaluation of the quadratic (see sphere_span)
            if constexpr (C::has(F_SPHERE) || C::has(F_MOVING)) {
              Ray<R> moved = ray, local = ray;
              if constexpr (C::has(F_XFORM)) {
                if (ob.flags & (OBJ_TRANSLATE | OBJ_ROTATE_Y)) to_local(ob, ray, moved, local);
              }
              const V3<R> center = s.tag == SH_SPHERE ? V3<R>(s.p[0], s.p[1], s.p[2]) : moving_center(s, local.time);
              if (!sphere_span(center, local, s.tag == SH_SPHERE ? s.p[3] : s.p[6], t1, t2)) return false;
            } else {
              return false;
            }
          } else if (!boundary_t<R, C>(ob, s, ray, ri, -inf, inf, -2, t1, unused)) {
            return false;
          }
          R t_after = t1 + (R)0.0001;
          if constexpr (Guard<R>::on) {
            if (!(t_after > t1)) t_after = nextafterf(t1, Lim<float>::inf());  // see object_hit
          }
          if (ball) { if (t2 < t_after) return false; }
          else if (!boundary_t<R, C>(ob, s, ray, ri, t_after, inf, -2, t2, unused)) return false;
          if (t1 < t_min) t1 = t_min;
          if (t2 > t_max) t2 = t_max;
          if (t1 >= t2) return false;
          if (t1 < (R)0) t1 = (R)0;
          R ray_length = length(ray.d);
          R distance_inside_boundary = (t2 - t1) * ray_length;
          R hit_distance = ob.neg_inv_density * gen_log01(media, medium_slot((uint32_t)s.obj_id, visit), (R)0);
          if (hit_distance > distance_inside_boundary) return false;
          t_out = t1 + hit_distance / ray_length;
          side_out = SIDE_MEDIUM;
          return true;
        }
      }
      return boundary_t<R, C>(ob, s, ray, ri, t_min, t_max, self_side, t_out, side_out);
    }
  }
  return shape_t<R, C>(s, ray, ri, t_min, t_max, self_side, t_out, side_out);
}

// The HitRecord of candidate (t, leaf, side), gathered per lane.
template <class R, class C> RT_DEV void finalize_hit(const SceneView<R, C>& sv, const Ray<R>& ray, R t, int leaf, int side, Rec<R>& rec) {
  const DShape<R> s = sv.shapes_lane[leaf];
  rec.t = t;
  rec.obj = s.obj_id;
  rec.sub = s.sub_id;
  rec.side = side;
  rec.leaf = leaf;
  rec.uv_lazy = 0;
  DObject<R> ob;
  ob.flags = 0u;
  if constexpr (C::has(F_XFORM) || C::has(F_MEDIUM)) ob = sv.objects_lane[leaf];
  if constexpr (C::has(F_MEDIUM)) {
    if (side == SIDE_MEDIUM) {  // src/constant_medium.rs:97-108
      rec.p = ray.at(t);
      rec.n = V3<R>((R)1, (R)0, (R)0);
      rec.front = 1;
      rec.mat = ob.medium_material;
      rec.side = -1;
      rec.leaf = -1;
      return;
    }
  }
  rec.mat = s.material;
  Ray<R> moved = ray, local = ray;
  if constexpr (C::has(F_XFORM)) to_local(ob, ray, moved, local);
  if (s.tag == SH_SPHERE || s.tag == SH_MOVING) {
    if constexpr (C::has(F_SPHERE) || C::has(F_MOVING)) {  // src/sphere.rs:64-70, src/moving_sphere.rs:64-69
      V3<R> center = s.tag == SH_SPHERE ? V3<R>(s.p[0], s.p[1], s.p[2]) : moving_center(s, local.time);
      R radius = s.tag == SH_SPHERE ? s.p[3] : s.p[6];
      V3<R> p = local.at(t);
      V3<R> outward = vdiv(p - center, radius);
      rec.p = p;
      rec.set_face_normal(local.d, outward);
      if constexpr (C::has(F_XFORM)) {
        if (ob.flags & OBJ_FRONT_CHAIN) {
          // A sphere re-centred from under Translate(RotateY(..)): the normal ends up against the ray either way, but front_face
          // is what RotateY's call left - the rotated normal against the direction in the UNrotated frame (src/hittable.rs:236-239);
          // Translate's call (src/hittable.rs:146-148) then reports whether that one already faced the ray.
          const R dlx = ob.cos_theta * ray.d.x - ob.sin_theta * ray.d.z, dlz = ob.sin_theta * ray.d.x + ob.cos_theta * ray.d.z;
          const R mixed = dlx * rec.n.x + ray.d.y * rec.n.y + dlz * rec.n.z;
          rec.front = mixed < (R)0;
        }
      }
      if constexpr (C::has(F_UV)) {
        if (s.tag == SH_SPHERE) { rec.u = outward.x; rec.v = outward.y; rec.uw = outward.z; rec.uv_lazy = 1; }
      }
    }
  } else {
    // plane of a rectangle or of a box side: constant axis and its coordinate k
    int axis;
    R k;
    if (s.tag == SH_BOX) {
      axis = side < 2 ? 2 : (side < 4 ? 1 : 0);
      k = side == 0 ? s.p[5] : side == 1 ? s.p[2] : side == 2 ? s.p[4] : side == 3 ? s.p[1] : side == 4 ? s.p[3] : s.p[0];
    } else {
      axis = s.tag == SH_RECT_XY ? 2 : (s.tag == SH_RECT_XZ ? 1 : 0);
      k = s.p[4];
    }
    if constexpr (C::has(F_UV)) {  // src/aarect.rs:93-100
      R oa = axis == 0 ? local.o.y : local.o.x, da = axis == 0 ? local.d.y : local.d.x;
      R ob2 = axis == 2 ? local.o.y : local.o.z, db = axis == 2 ? local.d.y : local.d.z;
      R a = oa + t * da, b = ob2 + t * db;
      R a0 = s.p[0], a1 = s.p[1], b0 = s.p[2], b1 = s.p[3];
      if (s.tag == SH_BOX) {  // extents of the side rectangle, src/box_struct.rs:22-71
        a0 = axis == 0 ? s.p[1] : s.p[0];
        a1 = axis == 0 ? s.p[4] : s.p[3];
        b0 = axis == 2 ? s.p[1] : s.p[2];
        b1 = axis == 2 ? s.p[4] : s.p[5];
      }
      rec.u = (a - a0) / (a1 - a0);
      rec.v = (b - b0) / (b1 - b0);
    }
    V3<R> outward(axis == 0 ? (R)1 : (R)0, axis == 1 ? (R)1 : (R)0, axis == 2 ? (R)1 : (R)0);
    rec.set_face_normal(local.d, outward);
    V3<R> p = local.at(t);
    if constexpr (Guard<R>::on) {  // see rect_hit
      if (axis == 0) p.x = k;
      if (axis == 1) p.y = k;
      if (axis == 2) p.z = k;
    }
    rec.p = p;
  }
  if constexpr (C::has(F_XFORM)) {
    if (ob.flags & OBJ_ROTATE_Y) {  // src/hittable.rs:229-239
      V3<R> p = rec.p, n = rec.n;
      p.x = ob.cos_theta * rec.p.x + ob.sin_theta * rec.p.z;
      p.z = -ob.sin_theta * rec.p.x + ob.cos_theta * rec.p.z;
      n.x = ob.cos_theta * rec.n.x + ob.sin_theta * rec.n.z;
      n.z = -ob.sin_theta * rec.n.x + ob.cos_theta * rec.n.z;
      rec.p = p;
      rec.set_face_normal(local.d, n);
    }
    if (ob.flags & OBJ_TRANSLATE) {  // src/hittable.rs:146-148
      rec.p = rec.p + V3<R>(ob.offset[0], ob.offset[1], ob.offset[2]);
      rec.set_face_normal(moved.d, rec.n);
    }
  }
}

// ---- forms used by the flat walk ----
// x in [lo, hi]; a NaN is outside.  Single precision: one median and one compare.
RT_DEV bool in_closed(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi) == x; }
RT_DEV bool in_closed(double x, double lo, double hi) { return (x >= lo) & (x <= hi); }

// Reciprocal of a direction component for the flat walk.  Single precision: the hardware estimate (1 ulp).  Double precision: the
// hardware estimate and one Newton step (0.6 ulp on average and 11 at most from the quotient 1/d that Aabb::hit multiplies by,
// src/aabb.rs:33); a plane distance formed with it is within a dozen ulp of the reference's quotient (k - o) / d.
RT_DEV float flat_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#ifdef RT_EXP_IEEE_RCP
RT_DEV double flat_rcp(double x) { return 1.0 / x; }
#else
RT_DEV double flat_rcp(double x) { return rcp_one_step(x); }
#endif
RT_DEV float next_up(float x) { return nextafterf(x, Lim<float>::inf()); }
RT_DEV double next_up(double x) { return nextafter(x, Lim<double>::inf()); }

// Distances along the ray as (plane - origin) * (1 / direction) in both precisions: the flat walk forms every plane distance the
// way the reference's box test does (src/aabb.rs:33-35).  Against the quotient (k - o) / d of src/aarect.rs:89 that is one more
// rounding and the reciprocal's own error (double precision: a dozen ulp at most), parts in 10^15 of t.
template <class R> struct RayRcp {
  V3<R> inv;
  RT_DEV explicit RayRcp(V3<R> dir) : inv(flat_rcp(dir.x), flat_rcp(dir.y), flat_rcp(dir.z)) {}
  RT_DEV RayRcp(V3<R> reciprocals, int) : inv(reciprocals) {}  // from reciprocals the caller already has
  template <int AXIS> RT_DEV R quot(R num) const { return num * (AXIS == 0 ? inv.x : AXIS == 1 ? inv.y : inv.z); }
};

template <int AXIS>
RT_DEV bool rect_fast(float a0, float a1, float b0, float b1, float k, const Ray<float>& ray, const RayRcp<float>& ri, float t_min, float t_max,
                      float& t_out) {
  constexpr int A = (AXIS == 0) ? 1 : 0;
  constexpr int B = (AXIS == 2) ? 1 : 2;
  const float o[3] = {ray.o.x, ray.o.y, ray.o.z};
  const float d[3] = {ray.d.x, ray.d.y, ray.d.z};
  const float t = ri.template quot<AXIS>(k - o[AXIS]);
  const float a = o[A] + t * d[A];
  const float b = o[B] + t * d[B];
  t_out = t;
  return in_closed(t, t_min, t_max) & in_closed(a, a0, a1) & in_closed(b, b0, b1);
}
// Double precision keeps the reference's six comparisons as written (negated, so that a NaN passes them as it does there).
template <int AXIS>
RT_DEV bool rect_fast(double a0, double a1, double b0, double b1, double k, const Ray<double>& ray, const RayRcp<double>& ri, double t_min,
                      double t_max, double& t_out) {
  return rect_t<double, AXIS, RayRcp<double>, false>(a0, a1, b0, b1, k, ray, ri, t_min, t_max, t_out);
}

// The six sides of a box (src/box_struct.rs:22-71) as three slabs: the ray meets the surface where it enters (largest of the
// per-axis near distances) and where it leaves (smallest far distance), provided enter <= leave; the closest of the two inside
// [t_min, t_max] is what testing the six rectangles one by one returns.  Side numbering and tie order (a later side wins:
// x sides over y sides over z sides) follow the reference's list.  `self_side` (>= 0) removes the side the ray starts on.
template <class R>
RT_DEV bool box_slab_t(const DShape<R>& s, const Ray<R>& ray, const RayRcp<R>& ri, R t_min, R t_max, int self_side, R& t_out, int& side_out) {
  const R x0 = (s.p[0] - ray.o.x) * ri.inv.x, x1 = (s.p[3] - ray.o.x) * ri.inv.x;
  const R y0 = (s.p[1] - ray.o.y) * ri.inv.y, y1 = (s.p[4] - ray.o.y) * ri.inv.y;
  const R z0 = (s.p[2] - ray.o.z) * ri.inv.z, z1 = (s.p[5] - ray.o.z) * ri.inv.z;
  const R nx = min_plain(x0, x1), fx = max_plain(x0, x1);
  const R ny = min_plain(y0, y1), fy = max_plain(y0, y1);
  const R nz = min_plain(z0, z1), fz = max_plain(z0, z1);
  const R tn = max_plain(max_plain(nx, ny), nz);
  const R tf = min_plain(min_plain(fx, fy), fz);
  // sides: 0 z=hi, 1 z=lo, 2 y=hi, 3 y=lo, 4 x=hi, 5 x=lo; the near plane of an axis is lo when the ray moves up that axis, and
  // its far plane is the other one of the pair (the code with its lowest bit flipped)
  const int cx = ray.d.x > (R)0 ? 5 : 4, cy = ray.d.y > (R)0 ? 3 : 2, cz = ray.d.z > (R)0 ? 1 : 0;
  const int near_side = nx == tn ? cx : (ny == tn ? cy : cz);
  const int far_side = (fx == tf ? cx : (fy == tf ? cy : cz)) ^ 1;
  const bool cross = tn <= tf;
  const bool enter_ok = cross & in_closed(tn, t_min, t_max) & (near_side != self_side);
  const bool leave_ok = cross & in_closed(tf, t_min, t_max) & (far_side != self_side);
  t_out = enter_ok ? tn : tf;
  side_out = enter_ok ? near_side : far_side;
  return enter_ok | leave_ok;
}

// Entry and exit distances of the line through a box (no range applied); false when the line misses it.
template <class R> RT_DEV bool box_slab_span(const DShape<R>& s, const Ray<R>& ray, const RayRcp<R>& ri, R& tn, R& tf) {
  const R x0 = (s.p[0] - ray.o.x) * ri.inv.x, x1 = (s.p[3] - ray.o.x) * ri.inv.x;
  const R y0 = (s.p[1] - ray.o.y) * ri.inv.y, y1 = (s.p[4] - ray.o.y) * ri.inv.y;
  const R z0 = (s.p[2] - ray.o.z) * ri.inv.z, z1 = (s.p[5] - ray.o.z) * ri.inv.z;
  tn = max_plain(max_plain(min_plain(x0, x1), min_plain(y0, y1)), min_plain(z0, z1));
  tf = min_plain(min_plain(max_plain(x0, x1), max_plain(y0, y1)), max_plain(z0, z1));
  return tn <= tf;
}

// Scenes of a few leaves: every leaf is tested with the same shrinking t_max and no box test in between.  A box test can only
// skip leaves the ray does not hit (up to rounding at grazing incidence), and with incoherent rays every wave visits every entry
// anyway, so the inner box tests and the per-lane position in the tree are pure overhead here.
//   Double precision: the leaves in the order the tree visits them, so that among exactly coincident surfaces (a box standing on
//   the floor) the same one is reported as by the reference's walk; a box is three slabs, and plane distances are
//   (plane - origin) * (1 / direction) (see RayRcp), so t is within a dozen ulp of the reference's quotient.
//   Single precision: plain rectangles first, grouped by axis, then the other leaves in visiting order.
template <class R, class C>
RT_DEV bool traverse_flat(const SceneView<R, C>& sv, int start, const Ray<R>& ray, R t_min, R t_max, Rec<R>& rec, Stream& media, Origin org) {
  const RayRcp<R> ri(ray.d);
  R closest = t_max;
  // the winning candidate so far as one word, (leaf << 3) | (side + 1), -1 = none: one select per accepted test instead of two
  int best = -1;
  // No box test at all, not even the root's: a ray that starts on a surface starts inside the root box, and camera rays that miss
  // it share their wave with lanes at other bounces, so the early out would never be taken by a whole wave.
  (void)start;
  RT_STAMP_IN(media, 3);
  {
    const int org_leaf = Guard<R>::on ? org.leaf() : -1, org_side = org.side();
    if constexpr (Guard<R>::on) {
      // plain rectangles, one list per constant axis: one record fetch and a dozen vector instructions each
      const FlatRect* fr = sv.frect;
      auto rect_list = [&](auto axis_tag, uint32_t n) {
        constexpr int AXIS = decltype(axis_tag)::value;
        for (uint32_t k = 0; k < n; ++k, ++fr) {
          const FlatRect r = load_uniform(fr);
          R t;
          const bool ok = rect_fast<AXIS>(r.a0, r.a1, r.b0, r.b1, r.k, ray, ri, t_min, closest, t) & (org_leaf != r.obj);
          closest = ok ? t : closest; best = ok ? (r.obj << 3) : best;
        }
      };
      rect_list(std::integral_constant<int, 0>{}, sv.n_frect[0]);
      rect_list(std::integral_constant<int, 1>{}, sv.n_frect[1]);
      rect_list(std::integral_constant<int, 2>{}, sv.n_frect[2]);
    }
    for (uint32_t k = 0; k < sv.n_flat; ++k) {
      const FlatLeaf<R> fl = load_uniform(sv.flat + k);
      const uint32_t word = fl.word;
      const int obj = (int)(word & LEAF_INDEX_MASK);
      DShape<R> s;
      s.tag = fl.tag; s.obj_id = fl.obj_id; s.material = 0; s.sub_id = 0;
#pragma unroll
      for (int q = 0; q < 10; ++q) s.p[q] = fl.p[q];
      const bool self = org_leaf == obj;
      R t = (R)0;
      int side = -1;
      bool ok = false;
      bool plain = (word & LEAF_SIMPLE) != 0u;
      Ray<R> local = ray;
      RayRcp<R> rl = ri;
      if constexpr (C::has(F_XFORM) || C::has(F_MEDIUM)) {
        if (!plain) {
          DObject<R> ob;
          ob.flags = fl.flags; ob.sub_root = -1; ob.medium_material = -1;
          ob.offset[0] = fl.offset[0]; ob.offset[1] = fl.offset[1]; ob.offset[2] = fl.offset[2];
          ob.sin_theta = fl.sin_theta; ob.cos_theta = fl.cos_theta; ob.neg_inv_density = fl.neg_inv_density;
          bool medium = false;
          if constexpr (C::has(F_MEDIUM)) medium = (ob.flags & OBJ_MEDIUM) != 0u;
          if (medium) {
            // src/constant_medium.rs:41-111.  The first crossing of the whole line is where the ray enters the boundary, the
            // first one at least 0.0001 later is where it leaves it: a box gives both from one slab evaluation, a sphere from
            // the two roots of its quadratic.
            Ray<R> moved;
            to_local(ob, ray, moved, local);
            if (ob.flags & OBJ_ROTATE_Y) {
              rl.inv.x = flat_rcp(local.d.x);
              rl.inv.z = flat_rcp(local.d.z);
            }
            R tn = (R)0, tf = (R)0;
            bool cross = false, spanned = false;
            if (s.tag == SH_BOX) { cross = box_slab_span(s, local, rl, tn, tf); spanned = true; }
            if (spanned) {
              R t_after = tn + (R)0.0001;
              if (!(t_after > tn)) t_after = next_up(tn);
              R t1 = tn, t2 = tf;
              bool in = cross & (tf >= t_after);
              if (t1 < t_min) t1 = t_min;
              if (t2 > closest) t2 = closest;
              in = in & !(t1 >= t2);
              if (t1 < (R)0) t1 = (R)0;
              if (in) {
                const R ray_length = length(ray.d);
                const R distance_inside_boundary = (t2 - t1) * ray_length;
                const R hit_distance = ob.neg_inv_density * gen_log01(media, medium_slot((uint32_t)s.obj_id, visit_bit(word, 0)), (R)0);
                if (!(hit_distance > distance_inside_boundary)) {
                  t = t1 + hit_distance / ray_length;
                  side = SIDE_MEDIUM;
                  ok = true;
                }
              }
            } else {
              const RayInv<R> rq(ray.d);
              ok = leaf_t<R, C>(sv, word, ray, rq, t_min, closest, media, org, t, side, visit_bit(word, 0));
            }
          } else {
            Ray<R> moved;
            to_local(ob, ray, moved, local);
            if (ob.flags & OBJ_ROTATE_Y) {  // the y component of the direction is unchanged
              rl.inv.x = flat_rcp(local.d.x);
              rl.inv.z = flat_rcp(local.d.z);
            }
            plain = true;
          }
        }
      }
      // the shape test proper.  Double precision: on the world ray for a bare shape and on the holder's local ray otherwise, as two
      // calls, so that the bare shapes - most leaves - do not pay for copies of the ray into the registers the transformed ones use
      auto shape_test = [&](const Ray<R>& r, const RayRcp<R>& rr) {
        switch (s.tag) {
          case SH_SPHERE: if constexpr (C::has(F_SPHERE)) {
            ok = sphere_roots(V3<R>(s.p[0], s.p[1], s.p[2]), r, s.p[3], t_min, closest, t, self);
          } break;
          case SH_MOVING: if constexpr (C::has(F_MOVING)) {
            ok = sphere_roots(moving_center(s, r.time), r, s.p[6], t_min, closest, t, self);
          } break;
          case SH_RECT_XY: if constexpr (C::has(F_RECT)) { ok = rect_fast<2>(s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], r, rr, t_min, closest, t) & !self; } break;
          case SH_RECT_XZ: if constexpr (C::has(F_RECT)) { ok = rect_fast<1>(s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], r, rr, t_min, closest, t) & !self; } break;
          case SH_RECT_YZ: if constexpr (C::has(F_RECT)) { ok = rect_fast<0>(s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], r, rr, t_min, closest, t) & !self; } break;
          case SH_BOX: if constexpr (C::has(F_BOX)) { ok = box_slab_t<R>(s, r, rr, t_min, closest, self ? org_side : -2, t, side); } break;
          default: break;
        }
      };
      if constexpr (sizeof(R) == sizeof(double)) {  // measured: +1.4 % on the headline frame in double, -2.5 % in single (one call there)
        if (word & LEAF_SIMPLE) shape_test(ray, ri);
        else if (plain) shape_test(local, rl);
      } else {
        if (plain) shape_test(local, rl);
      }
      closest = ok ? t : closest;
      best = ok ? ((obj << 3) | (side + 1)) : best;
    }
  }
  RT_STAMP_IN(media, 4);
  if (best < 0) return false;
  const int best_leaf = best >> 3, best_side = (best & 7) - 1;
  if constexpr (!C::has(F_UV) && !C::has(F_SPHERE) && !C::has(F_MOVING)) {
    // every surface here is a planar side: its record comes from the table of sides.  The normal goes through the reference's
    // chain of set_face_normal calls - shape, RotateY (which pairs the rotated normal with the UNrotated-frame direction,
    // src/hittable.rs:236-239), Translate (src/hittable.rs:146-148) - as signs of one vector.
    if (best_side != SIDE_MEDIUM) {
      // record addresses by 24-bit multiplies (full rate; a 32-bit integer multiply takes four times as long)
      const uint32_t face_off = __umul24((uint32_t)best_leaf, 6u * (uint32_t)sizeof(FlatFace<R>)) + (uint32_t)(best_side < 0 ? 0 : best_side) * (uint32_t)sizeof(FlatFace<R>);
      const FlatFace<R> f = *reinterpret_cast<const FlatFace<R>*>(reinterpret_cast<const uint8_t*>(sv.faces_lane) + face_off);
      const DShape<R>& s = *reinterpret_cast<const DShape<R>*>(reinterpret_cast<const uint8_t*>(sv.shapes_lane) + __umul24((uint32_t)best_leaf, (uint32_t)sizeof(DShape<R>)));
      const int axis = (int)((f.flags >> 4) & 3u);
      const V3<R> nw(f.n[0], f.n[1], f.n[2]);
      V3<R> dl = ray.d;
      if (f.flags & OBJ_ROTATE_Y) {
        dl.x = f.cos_theta * ray.d.x - f.sin_theta * ray.d.z;
        dl.z = f.sin_theta * ray.d.x + f.cos_theta * ray.d.z;
      }
      const R d1 = axis == 0 ? dl.x : axis == 1 ? dl.y : dl.z;
      bool front = d1 < (R)0;
      R sgn = front ? (R)1 : (R)-1;
      if (f.flags & OBJ_ROTATE_Y) { front = sgn * dot(dl, nw) < (R)0; sgn = front ? sgn : -sgn; }
      if (f.flags & OBJ_TRANSLATE) { front = sgn * dot(ray.d, nw) < (R)0; sgn = front ? sgn : -sgn; }
      V3<R> p = ray.at(closest);
      if constexpr (Guard<R>::on) {
        if (!(f.flags & (OBJ_ROTATE_Y | OBJ_TRANSLATE))) {  // an untransformed side: the point lies on its plane exactly
          p.x = axis == 0 ? f.k : p.x; p.y = axis == 1 ? f.k : p.y; p.z = axis == 2 ? f.k : p.z;
        }
      }
      rec.t = closest; rec.p = p; rec.n = sgn * nw; rec.front = front;
      rec.mat = s.material; rec.obj = s.obj_id; rec.sub = s.sub_id;
      rec.side = best_side; rec.leaf = best_leaf; rec.uv_lazy = 0;
      RT_STAMP_IN(media, 5);
      return true;
    }
  }
  finalize_hit<R, C>(sv, ray, closest, best_leaf, best_side, rec);
  RT_STAMP_IN(media, 5);
  return true;
}

// Single precision, per-lane walk with the record deferred: the walk keeps (t, leaf, side, holder of the nested tree) and builds
// the HitRecord once.  Apart from using far fewer registers than carrying a full record through two tree levels, it performs
// the tests of traverse_lane.  (The stale (u, v) a moving sphere or a medium inherits in the reference is not reproduced here;
// the double-precision kernels, which are the ones compared field by field, keep the eager walk.)
template <class C>
RT_DEV bool traverse_lane_deferred(const SceneView<float, C>& sv, int start, const Ray<float>& ray, float t_min, float t_max, Rec<float>& rec,
                                   Stream& media, Origin org, Visits* tally = nullptr) {
  const LaneBoxTest<float> box(ray);
  RayInv<float> ri(ray.d);
  ri.inv = box.inv;
  float closest = t_max;
  int best_leaf = -1, best_side = -1, best_holder = -1;
  int i = start;
  while (i >= 0) {
    DLink<float> lk = sv.link(i);
    while (!(lk.hit & LEAF_FLAG)) {
      if (tally) ++tally->boxes;
      i = box.hit(lk, ray, t_min, closest, sv.exact_boxes, i) ? (int)lk.hit : lk.miss;
      if (i < 0) break;
      lk = sv.link(i);
    }
    if (i < 0) break;
    if (tally) ++tally->leaves;
    const int obj = (int)(lk.hit & LEAF_INDEX_MASK);
    bool nested = false;
    if constexpr (C::has(F_NESTED)) {
      if (lk.hit & LEAF_NESTED) {
        nested = true;
        const DObject<float> ob = sv.object(obj);
        Ray<float> moved, local;
        to_local(ob, ray, moved, local);
        const LaneBoxTest<float> lbox(local);
        RayInv<float> rl(local.d);
        rl.inv = lbox.inv;
        // closest candidate of the nested tree inside [lo, hi] (BVHNode::hit on the holder's tree, src/bvh.rs:108-119)
        auto nested_closest = [&](float lo, float hi, Origin from, float& t_best, int& leaf_best, int& side_best) {
          bool found = false;
          int j = ob.sub_root;
          while (j >= 0) {
            DLink<float> nl = sv.link(j);
            while (!(nl.hit & LEAF_FLAG)) {
              if (tally) ++tally->boxes;
              j = lbox.hit(nl, local, lo, hi, sv.exact_boxes, j) ? (int)nl.hit : nl.miss;
              if (j < 0) break;
              nl = sv.link(j);
            }
            if (j < 0) break;
            if (tally) ++tally->leaves;
            float t;
            int side;
            if (leaf_t<float, C>(sv, nl.hit, local, rl, lo, hi, media, from, t, side, visit_bit(lk.hit, 0) | visit_bit(nl.hit, 1))) {
              hi = t;
              t_best = t;
              leaf_best = (int)(nl.hit & LEAF_INDEX_MASK);
              side_best = side;
              found = true;
            }
            j = nl.miss;
          }
          return found;
        };
        bool medium = false;
        if constexpr (C::has(F_MEDIUM)) medium = (ob.flags & OBJ_MEDIUM) != 0u;
        if (medium) {
          // ConstantMedium whose boundary is the nested tree (src/constant_medium.rs:41-111): first crossing of the whole line,
          // first crossing at least 0.0001 later, then the free-path draw of the holder
          const float inf = Lim<float>::inf();
          float t1 = 0.0f, t2 = 0.0f;
          int l_unused = -1, s_unused = -1;
          if (nested_closest(-inf, inf, Origin(), t1, l_unused, s_unused)) {
            float t_after = t1 + 0.0001f;
            if (!(t_after > t1)) t_after = nextafterf(t1, inf);
            if (nested_closest(t_after, inf, Origin(), t2, l_unused, s_unused)) {
              if (t1 < t_min) t1 = t_min;
              if (t2 > closest) t2 = closest;
              if (!(t1 >= t2)) {
                if (t1 < 0.0f) t1 = 0.0f;
                const float ray_length = length(ray.d);
                const float distance_inside_boundary = (t2 - t1) * ray_length;
                const float hit_distance = ob.neg_inv_density * gen_log01(media, medium_slot((uint32_t)sv.shape(obj).obj_id, visit_bit(lk.hit, 0)), 0.0f);
                if (!(hit_distance > distance_inside_boundary)) {
                  closest = t1 + hit_distance / ray_length;
                  best_leaf = obj;
                  best_side = SIDE_MEDIUM;
                  best_holder = -1;
                }
              }
            }
          }
        } else {
          float t = 0.0f;
          int leaf = -1, side = -1;
          if (nested_closest(t_min, closest, org, t, leaf, side)) {
            closest = t;
            best_leaf = leaf;
            best_side = side;
            best_holder = obj;
          }
        }
      }
    }
    if (!nested) {
      float t;
      int side;
      if (leaf_t<float, C>(sv, lk.hit, ray, ri, t_min, closest, media, org, t, side, visit_bit(lk.hit, 0))) {
        closest = t;
        best_leaf = obj;
        best_side = side;
        best_holder = -1;
      }
    }
    i = lk.miss;
  }
  if (best_leaf < 0) return false;
  if constexpr (C::has(F_NESTED)) {
    if (best_holder >= 0) {  // record in the space of the nested tree, then back out through its holder (src/hittable.rs:146-148,229-239)
      const DObject<float> ob = sv.objects_lane[best_holder];
      Ray<float> moved, local;
      to_local(ob, ray, moved, local);
      finalize_hit<float, C>(sv, local, closest, best_leaf, best_side, rec);
      if (ob.flags & OBJ_ROTATE_Y) {
        V3<float> p = rec.p, n = rec.n;
        p.x = ob.cos_theta * rec.p.x + ob.sin_theta * rec.p.z;
        p.z = -ob.sin_theta * rec.p.x + ob.cos_theta * rec.p.z;
        n.x = ob.cos_theta * rec.n.x + ob.sin_theta * rec.n.z;
        n.z = -ob.sin_theta * rec.n.x + ob.cos_theta * rec.n.z;
        rec.p = p;
        rec.set_face_normal(local.d, n);
      }
      if (ob.flags & OBJ_TRANSLATE) {
        rec.p = rec.p + V3<float>(ob.offset[0], ob.offset[1], ob.offset[2]);
        rec.set_face_normal(moved.d, rec.n);
      }
      return true;
    }
  }
  finalize_hit<float, C>(sv, ray, closest, best_leaf, best_side, rec);
  return true;
}

// The four-wide walk of one ray from start to end (the path kernels step through it instead: k_paths_stepwise), records built as
// k_paths_stepwise builds them (Cfg::defer_records).
// `tie` (F_EXACT configurations): set when two different leaves report exactly the same distance, the one case in which the order
// of the tests decides what is reported (src/bvh.rs:108-119 gives it to the leaf its walk tests last, if its boxes let it get
// there); the caller then asks the list walk, which keeps the reference's order.
template <class R, class C>
RT_DEV bool traverse_wide(const SceneView<R, C>& sv, const Ray<R>& ray, R t_min, R t_max, Rec<R>& rec, Stream& media, Origin org, Visits* tally,
                          bool* tie = nullptr) {
  constexpr bool DEFER = C::template defer_records<R>();
  const LaneBoxTest<R> box(ray, sv.narrow_ok);
  WideLane<R> w;
  w.start();
  R closest = t_max;
  bool any = false;
  int best_leaf = -1, best_side = -1;
  auto test_leaf = [&](uint32_t word) {
    if (tally) ++tally->leaves;
    const int leaf = (int)(word & LEAF_INDEX_MASK);
    if constexpr (DEFER) {
      RayInv<R> ri(ray.d);
      ri.inv = box.inv;
      R t;
      int side;
      if (leaf_t<R, C>(sv, word, ray, ri, t_min, closest, media, org, t, side, visit_bit(word, 0))) {
        if (tie && best_leaf >= 0 && best_leaf != leaf && t == closest) *tie = true;
        closest = t; best_leaf = leaf; best_side = side;
      }
    } else {
      const int prev_leaf = rec.leaf;
      const R prev_t = closest;
      if (leaf_hit<R, C, 0>(sv, word, ray, t_min, closest, rec, media, org)) {
        if (tie && any && rec.leaf >= 0 && prev_leaf >= 0 && prev_leaf != rec.leaf && rec.t == prev_t) *tie = true;
        closest = rec.t; any = true;
      }
    }
  };
  for (uint32_t k = 0; k < sv.n_pre; ++k) test_leaf(sv.pre[k]);  // objects as large as the scene, kept outside the tree
  while (!w.idle()) {
    while (w.at_box()) {
      if (tally) tally->boxes += 4;  // four boxes, one 128-byte record
      w.box_step(sv.wide, sv.stack, box, ray, t_min, closest);
    }
    if (w.idle()) break;
    test_leaf(w.cur);
    w.pop(sv.stack);
  }
  if constexpr (DEFER) {
    if (best_leaf < 0) return false;
    finalize_hit<R, C>(sv, ray, closest, best_leaf, best_side, rec);
    return true;
  }
  return any;
}

// world.hit(ray, t_min, t_max, rec).  May be called from divergent code: lanes without a ray simply do not take part.
template <class R, class C>
RT_DEV bool closest_hit(const SceneView<R, C>& sv, int root, const Ray<R>& ray, R t_min, R t_max, Rec<R>& rec, Stream& media, Origin org,
                        Visits* tally = nullptr) {
  if constexpr (C::walk == WALK_UNIFORM) {
    if (tally) tally->leaves += sv.n_flat + sv.n_frect[0] + sv.n_frect[1] + sv.n_frect[2];  // every leaf, no box
    return traverse_flat<R, C>(sv, root, ray, t_min, t_max, rec, media, org);
  } else if constexpr (C::walk == WALK_LANE_WIDE) {
    if constexpr (C::has(F_EXACT) && sizeof(R) == sizeof(double)) {
      bool tie = false;
      const bool hit = traverse_wide<R, C>(sv, ray, t_min, t_max, rec, media, org, tally, &tie);
      if (!tie) return hit;
      rec = Rec<R>();
      return traverse_lane<R, C, 0>(sv, root, ray, t_min, t_max, rec, media, org, 0u, nullptr);
    } else {
      return traverse_wide<R, C>(sv, ray, t_min, t_max, rec, media, org, tally);
    }
  } else {
    if constexpr (sizeof(R) == sizeof(float)) return traverse_lane_deferred<C>(sv, root, ray, t_min, t_max, rec, media, org, tally);
    else return traverse_lane<R, C, 0>(sv, root, ray, t_min, t_max, rec, media, org, 0u, tally);
  }
}

}  // namespace rtd
