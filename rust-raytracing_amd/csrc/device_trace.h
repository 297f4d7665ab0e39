// Closest-hit query: world.hit(ray, t_min, t_max, rec) of the reference, for one ray per lane.
//   BVH descent        src/bvh.rs:108-119      -> traverse()
//   Aabb::hit          src/aabb.rs:30-46       -> aabb_hit()
//   Sphere             src/sphere.rs:38-73     MovingSphere  src/moving_sphere.rs:44-72
//   XY/XZ/YZRect       src/aarect.rs:88-177    Box           src/box_struct.rs:86-94 + src/hittable_list.rs:29-43
//   Translate/RotateY  src/hittable.rs:140-151,213-242
//   ConstantMedium     src/constant_medium.rs:41-111
#pragma once
#include "device_scene.h"

namespace rtd {

template <class R> struct Ray {
  V3<R> o, d;
  R time;
  RT_DEV V3<R> at(R t) const { return o + (t * d); }  // src/ray.rs:31-33
};

// HitRecord (src/hittable.rs:10-18).  `uv_lazy`: (u, v, uw) hold the unit outward normal of the last accepted
// static sphere and get_sphere_uv (src/sphere.rs:24-34) is evaluated once, when the record is final, instead
// of on every accepted candidate; a record written by a shape that leaves u, v untouched (moving sphere,
// medium) keeps whatever was there, exactly as the reference's shared `rec` does.
template <class R> struct Rec {
  V3<R> p, n;
  R t, u, v, uw;
  int32_t front, mat, obj, sub, side, uv_lazy;
  int32_t leaf;  // device index of the object whose surface was hit, -1 for a medium (see Origin)
  RT_DEV Rec() : t(0), u(0), v(0), uw(0), front(0), mat(-1), obj(-1), sub(-1), side(-1), uv_lazy(0), leaf(-1) {}
  RT_DEV void set_face_normal(V3<R> dir, V3<R> outward) {  // src/hittable.rs:79-86
    front = dot(dir, outward) < (R)0;
    n = front ? outward : -outward;
  }
  RT_DEV void finish_uv() {  // src/sphere.rs:24-34
    if (uv_lazy) {
      const R PI = (R)3.14159265358979323846264338327950288;
      R theta = rt_acos(-v);
      R phi = rt_atan2(-uw, u) + PI;
      u = phi / ((R)2 * PI);
      v = theta / PI;
      uv_lazy = 0;
    }
  }
};

// Surface a ray starts on, packed as (leaf + 1) << 3 | (side + 1); 0 = none (camera rays, medium scatter, test rays).
//
// Single precision only.  The reference separates a scattered ray from its own surface with t_min = 0.001 measured in
// units of |direction|; a Lambertian direction can be 1e-3 long, so the physical gap is as small as 1e-6, far below the
// f32 rounding error of a hit point at coordinates of a few hundred (3e-5), and the ray would re-hit its own surface and
// stay trapped behind it.  In exact arithmetic a ray leaving a plane meets that plane again only at t = 0 and a ray
// leaving a sphere meets it at t = 0 and t = -2*half_b/a (the constant term of the quadratic is exactly 0); the f32
// kernels use exactly those facts for the one primitive the ray starts on, which is what the f64 evaluation reproduces
// to 1e-13.  The f64 kernels never look at the origin tag.
struct Origin {
  uint32_t bits;
  RT_DEV Origin() : bits(0) {}
  RT_DEV explicit Origin(uint32_t b) : bits(b) {}
  RT_DEV Origin(int leaf, int side) : bits(leaf < 0 ? 0u : (((uint32_t)leaf + 1u) << 3) | (uint32_t)(side + 1)) {}
  RT_DEV int leaf() const { return (int)(bits >> 3) - 1; }
  RT_DEV int side() const { return (int)(bits & 7u) - 1; }
};
// Diagnostic tally of one closest-hit query on a per-lane walk: box entries tested and leaves tested (what the memory roofline
// of the large-scene kernels is priced in).  Only the inspection kernel's counting variant passes one; everywhere else the
// pointer is a compile-time null and the increments vanish.
struct Visits {
  uint32_t boxes = 0, leaves = 0;
};
template <class R> struct Guard { static constexpr bool on = false; };
template <> struct Guard<float> { static constexpr bool on = true; };

// src/aabb.rs:30-46 with the three reciprocals hoisted out (the reference recomputes 1/d per box; same values).
template <class R> RT_DEV bool aabb_hit(const DLink<R>& b, const Ray<R>& ray, V3<R> inv_d, R t_min, R t_max) {
  bool ok = true;
  {
    R t0 = (b.bmin[0] - ray.o.x) * inv_d.x, t1 = (b.bmax[0] - ray.o.x) * inv_d.x;
    if (inv_d.x < (R)0) { R s = t0; t0 = t1; t1 = s; }
    t_min = t0 > t_min ? t0 : t_min;
    t_max = t1 < t_max ? t1 : t_max;
    ok = ok && !(t_max <= t_min);
  }
  {
    R t0 = (b.bmin[1] - ray.o.y) * inv_d.y, t1 = (b.bmax[1] - ray.o.y) * inv_d.y;
    if (inv_d.y < (R)0) { R s = t0; t0 = t1; t1 = s; }
    t_min = t0 > t_min ? t0 : t_min;
    t_max = t1 < t_max ? t1 : t_max;
    ok = ok && !(t_max <= t_min);
  }
  {
    R t0 = (b.bmin[2] - ray.o.z) * inv_d.z, t1 = (b.bmax[2] - ray.o.z) * inv_d.z;
    if (inv_d.z < (R)0) { R s = t0; t0 = t1; t1 = s; }
    t_min = t0 > t_min ? t0 : t_min;
    t_max = t1 < t_max ? t1 : t_max;
    ok = ok && !(t_max <= t_min);
  }
  return ok;
}

// Quadratic of src/sphere.rs:45-62 for a sphere at `center`.  Double precision: the reference's expressions as written.
RT_DEV bool sphere_roots(V3<double> center, const Ray<double>& ray, double radius, double t_min, double t_max, double& root, bool) {
  // never contracted into multiply-adds, whatever the translation unit's setting: `half_b^2 - a*c` cancels on the radius-1000
  // ground spheres, and a differently rounded product there shows in t at 1e-11
#pragma clang fp contract(off)
  V3<double> oc = ray.o - center;
  double a = length_squared(ray.d);
  double half_b = dot(oc, ray.d);
  double c = length_squared(oc) - radius * radius;
  double discriminant = half_b * half_b - a * c;
  if (discriminant < 0.0) return false;
  double sqrt_d = rt_sqrt(discriminant);
  root = (-half_b - sqrt_d) / a;
  if (root < t_min || t_max < root) {
    root = (-half_b + sqrt_d) / a;
    if (root < t_min || t_max < root) return false;
  }
  return true;
}
// Single precision: `half_b^2 - a*c` cancels catastrophically for the radius-1000 ground spheres and for spheres far from
// the ray origin, so oc, the three coefficients, the discriminant and the numerator -half_b -/+ sqrt(D) are formed in f64
// from the f32 inputs (products of floats are exact in double); the square root and the division (a reciprocal) stay in f32.
RT_DEV bool sphere_roots(V3<float> center, const Ray<float>& ray, float radius, float t_min, float t_max, float& root, bool on_surface) {
  double ocx = (double)ray.o.x - (double)center.x, ocy = (double)ray.o.y - (double)center.y, ocz = (double)ray.o.z - (double)center.z;
  double dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
  double a = dx * dx + dy * dy + dz * dz;
  double half_b = ocx * dx + ocy * dy + ocz * dz;
  double c = ocx * ocx + ocy * ocy + ocz * ocz - (double)radius * (double)radius;
  if (on_surface) c = 0.0;  // the ray starts on this sphere: see Origin
  double discriminant = half_b * half_b - a * c;
  if (discriminant < 0.0) return false;
  const float inv_a = __builtin_amdgcn_rcpf((float)a);  // hardware reciprocal, 1 ulp
  if (on_surface) {
    // c = 0: the roots are 0 (the surface the ray stands on: never a hit) and -2 half_b / a, formed without a square root so
    // that no rounding of it can push the zero root past t_min (on a sphere of radius 1e5 it would)
    root = (float)(-2.0 * half_b) * inv_a;
    return !(root < t_min) & !(t_max < root);
  }
  // the square root in single precision, except for spheres so large that its rounding would be visible in t
  const double sqrt_d = radius * radius > 1.0e4f ? __builtin_sqrt(discriminant) : (double)__builtin_amdgcn_sqrtf((float)discriminant);
  const float near = (float)(-half_b - sqrt_d) * inv_a;
  const float far = (float)(-half_b + sqrt_d) * inv_a;
  const bool near_ok = !(near < t_min) & !(t_max < near);
  const bool far_ok = !(far < t_min) & !(t_max < far);
  root = near_ok ? near : far;
  return near_ok | far_ok;
}

// Both roots of the same quadratic, formed exactly as sphere_roots forms them (no range applied); false when the line misses the
// sphere.  What a ConstantMedium asks of a spherical boundary - the first crossing of the whole line, then the first one at least
// 0.0001 later (src/constant_medium.rs:50-60) - is the smaller root and then the larger one: one evaluation instead of two.
RT_DEV bool sphere_span(V3<double> center, const Ray<double>& ray, double radius, double& near, double& far) {
#pragma clang fp contract(off)
  V3<double> oc = ray.o - center;
  double a = length_squared(ray.d);
  double half_b = dot(oc, ray.d);
  double c = length_squared(oc) - radius * radius;
  double discriminant = half_b * half_b - a * c;
  if (discriminant < 0.0) return false;
  double sqrt_d = rt_sqrt(discriminant);
  near = (-half_b - sqrt_d) / a;
  far = (-half_b + sqrt_d) / a;
  return true;
}
RT_DEV bool sphere_span(V3<float> center, const Ray<float>& ray, float radius, float& near, float& far) {
  double ocx = (double)ray.o.x - (double)center.x, ocy = (double)ray.o.y - (double)center.y, ocz = (double)ray.o.z - (double)center.z;
  double dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
  double a = dx * dx + dy * dy + dz * dz;
  double half_b = ocx * dx + ocy * dy + ocz * dz;
  double c = ocx * ocx + ocy * ocy + ocz * ocz - (double)radius * (double)radius;
  double discriminant = half_b * half_b - a * c;
  if (discriminant < 0.0) return false;
  const float inv_a = __builtin_amdgcn_rcpf((float)a);
  const double sqrt_d = radius * radius > 1.0e4f ? __builtin_sqrt(discriminant) : (double)__builtin_amdgcn_sqrtf((float)discriminant);
  near = (float)(-half_b - sqrt_d) * inv_a;
  far = (float)(-half_b + sqrt_d) * inv_a;
  return true;
}

// One axis-aligned rectangle.  AXIS is the constant axis; (A, B) the in-plane axes in the reference's (u, v) order.
template <class R, int AXIS, bool UV>
RT_DEV bool rect_hit(R a0, R a1, R b0, R b1, R k, int material, const Ray<R>& ray, R t_min, R t_max, Rec<R>& rec) {  // src/aarect.rs:88-107
  constexpr int A = (AXIS == 0) ? 1 : 0;
  constexpr int B = (AXIS == 2) ? 1 : 2;
  const R o[3] = {ray.o.x, ray.o.y, ray.o.z};
  const R d[3] = {ray.d.x, ray.d.y, ray.d.z};
  R t = (k - o[AXIS]) / d[AXIS];
  if (t < t_min || t > t_max) return false;
  R a = o[A] + t * d[A];
  R b = o[B] + t * d[B];
  if (a < a0 || a > a1 || b < b0 || b > b1) return false;
  if constexpr (UV) {  // no texture of the scene reads (u, v) otherwise
    rec.u = (a - a0) / (a1 - a0);
    rec.v = (b - b0) / (b1 - b0);
    rec.uv_lazy = 0;
  }
  rec.t = t;
  V3<R> outward(AXIS == 0 ? (R)1 : (R)0, AXIS == 1 ? (R)1 : (R)0, AXIS == 2 ? (R)1 : (R)0);
  rec.set_face_normal(ray.d, outward);
  rec.mat = material;
  rec.p = ray.at(t);
  if constexpr (Guard<R>::on) {  // the hit point lies in the plane: o[AXIS] + t*d[AXIS] is k up to rounding, f64 keeps 1e-13 of it, f32 does not
    if (AXIS == 0) rec.p.x = k;
    if (AXIS == 1) rec.p.y = k;
    if (AXIS == 2) rec.p.z = k;
  }
  return true;
}

RT_DEV V3<double> moving_center(const DShape<double>& s, double time) {  // src/moving_sphere.rs:37-40
  V3<double> c0(s.p[0], s.p[1], s.p[2]), c1(s.p[3], s.p[4], s.p[5]);
  return c0 + ((time - s.p[7]) / (s.p[8] - s.p[7])) * (c1 - c0);
}
RT_DEV V3<float> moving_center(const DShape<float>& s, float time) {  // the quotient by the hardware reciprocal (1 ulp)
  V3<float> c0(s.p[0], s.p[1], s.p[2]), c1(s.p[3], s.p[4], s.p[5]);
  return c0 + ((time - s.p[7]) * __builtin_amdgcn_rcpf(s.p[8] - s.p[7])) * (c1 - c0);
}

// `self_side`: -2 when the ray does not start on this object, otherwise the side it starts on (-1 for one-sided shapes).
template <class R, class C> RT_DEV bool shape_hit(const DShape<R>& s, int leaf, const Ray<R>& ray, R t_min, R t_max, Rec<R>& rec, int self_side) {
  constexpr bool UV = C::has(F_UV);
  bool hit = false;
  const bool self = Guard<R>::on && self_side != -2;
  switch (s.tag) {
    case SH_SPHERE: if constexpr (C::has(F_SPHERE)) {  // src/sphere.rs:38-73
      V3<R> center(s.p[0], s.p[1], s.p[2]);
      R radius = s.p[3], root;
      if (sphere_roots(center, ray, radius, t_min, t_max, root, self)) {
        V3<R> p = ray.at(root);
        V3<R> outward = vdiv(p - center, radius);
        rec.t = root;
        rec.p = p;
        rec.set_face_normal(ray.d, outward);
        rec.mat = s.material;
        if constexpr (UV) { rec.u = outward.x; rec.v = outward.y; rec.uw = outward.z; rec.uv_lazy = 1; }
        rec.side = -1;
        hit = true;
      }
    } break;
    case SH_MOVING: if constexpr (C::has(F_MOVING)) {  // src/moving_sphere.rs:44-72 (u, v are not written)
      V3<R> center = moving_center(s, ray.time);
      R radius = s.p[6], root;
      if (sphere_roots(center, ray, radius, t_min, t_max, root, self)) {
        V3<R> p = ray.at(root);
        V3<R> outward = vdiv(p - moving_center(s, ray.time), radius);
        rec.t = root;
        rec.p = p;
        rec.set_face_normal(ray.d, outward);
        rec.mat = s.material;
        rec.side = -1;
        hit = true;
      }
    } break;
    case SH_RECT_XY: if constexpr (C::has(F_RECT)) { hit = !self && rect_hit<R, 2, UV>(s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], s.material, ray, t_min, t_max, rec); if (hit) rec.side = -1; } break;
    case SH_RECT_XZ: if constexpr (C::has(F_RECT)) { hit = !self && rect_hit<R, 1, UV>(s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], s.material, ray, t_min, t_max, rec); if (hit) rec.side = -1; } break;
    case SH_RECT_YZ: if constexpr (C::has(F_RECT)) { hit = !self && rect_hit<R, 0, UV>(s.p[0], s.p[1], s.p[2], s.p[3], s.p[4], s.material, ray, t_min, t_max, rec); if (hit) rec.side = -1; } break;
    case SH_BOX: if constexpr (C::has(F_BOX)) {
      // Six sides in the order of src/box_struct.rs:22-71, closest wins, a later side wins a tie
      // (src/hittable_list.rs:29-43: each success shrinks closest_so_far and overwrites the record).
      R lox = s.p[0], loy = s.p[1], loz = s.p[2], hix = s.p[3], hiy = s.p[4], hiz = s.p[5];
      R closest = t_max;
      if (!(self && self_side == 0) && rect_hit<R, 2, UV>(lox, hix, loy, hiy, hiz, s.material, ray, t_min, closest, rec)) { closest = rec.t; rec.side = 0; hit = true; }
      if (!(self && self_side == 1) && rect_hit<R, 2, UV>(lox, hix, loy, hiy, loz, s.material, ray, t_min, closest, rec)) { closest = rec.t; rec.side = 1; hit = true; }
      if (!(self && self_side == 2) && rect_hit<R, 1, UV>(lox, hix, loz, hiz, hiy, s.material, ray, t_min, closest, rec)) { closest = rec.t; rec.side = 2; hit = true; }
      if (!(self && self_side == 3) && rect_hit<R, 1, UV>(lox, hix, loz, hiz, loy, s.material, ray, t_min, closest, rec)) { closest = rec.t; rec.side = 3; hit = true; }
      if (!(self && self_side == 4) && rect_hit<R, 0, UV>(loy, hiy, loz, hiz, hix, s.material, ray, t_min, closest, rec)) { closest = rec.t; rec.side = 4; hit = true; }
      if (!(self && self_side == 5) && rect_hit<R, 0, UV>(loy, hiy, loz, hiz, lox, s.material, ray, t_min, closest, rec)) { closest = rec.t; rec.side = 5; hit = true; }
    } break;
    default: break;
  }
  if (hit) { rec.obj = s.obj_id; rec.sub = s.sub_id; rec.leaf = leaf; }
  return hit;
}

template <class R, class C, int LEVEL>
RT_DEV bool traverse_lane(const SceneView<R, C>& sv, int start, const Ray<R>& ray, R t_min, R t_max, Rec<R>& rec, Stream& media, Origin org,
                          uint32_t visit = 0u, Visits* tally = nullptr);

// What RotateY::hit and Translate::hit do to the record of a hit found below them (src/hittable.rs:229-239, 146-148).
template <class R> RT_DEV void record_to_world(const DObject<R>& ob, V3<R> local_d, V3<R> moved_d, Rec<R>& rec) {
  if (ob.flags & OBJ_ROTATE_Y) {  // src/hittable.rs:229-239
    V3<R> p = rec.p, n = rec.n;
    p.x = ob.cos_theta * rec.p.x + ob.sin_theta * rec.p.z;
    p.z = -ob.sin_theta * rec.p.x + ob.cos_theta * rec.p.z;
    n.x = ob.cos_theta * rec.n.x + ob.sin_theta * rec.n.z;
    n.z = -ob.sin_theta * rec.n.x + ob.cos_theta * rec.n.z;
    rec.p = p;
    rec.set_face_normal(local_d, n);  // object-space direction against a world-space normal, as the reference does
  }
  if (ob.flags & OBJ_TRANSLATE) {  // src/hittable.rs:146-148
    rec.p = rec.p + V3<R>(ob.offset[0], ob.offset[1], ob.offset[2]);
    rec.set_face_normal(moved_d, rec.n);
  }
}

// [Translate(] [RotateY(] geometry [))] — the part of an object below an optional ConstantMedium.
template <class R, class C, int LEVEL>
RT_DEV bool boundary_hit(const SceneView<R, C>& sv, const DObject<R>& ob, int obj_index, const Ray<R>& ray, R t_min, R t_max, Rec<R>& rec,
                         Stream& media, Origin org, uint32_t visit) {
  Ray<R> moved = ray;
  Ray<R> local = ray;
  if constexpr (C::has(F_XFORM)) {
    if (ob.flags & OBJ_TRANSLATE) moved.o = ray.o - V3<R>(ob.offset[0], ob.offset[1], ob.offset[2]);  // src/hittable.rs:141
    local = moved;
    if (ob.flags & OBJ_ROTATE_Y) {  // src/hittable.rs:214-223 (ray time forced to 0)
      local.o.x = ob.cos_theta * moved.o.x - ob.sin_theta * moved.o.z;
      local.o.z = ob.sin_theta * moved.o.x + ob.cos_theta * moved.o.z;
      local.d.x = ob.cos_theta * moved.d.x - ob.sin_theta * moved.d.z;
      local.d.z = ob.sin_theta * moved.d.x + ob.cos_theta * moved.d.z;
      local.time = (R)0;
    }
  }
  bool hit = false;
  bool nested = false;
  if constexpr (C::has(F_NESTED) && LEVEL == 0) {  // nested trees do not nest further (rejected at upload)
    if (ob.flags & OBJ_SUB_BVH) {
      nested = true;
      hit = traverse_lane<R, C, 1>(sv, ob.sub_root, local, t_min, t_max, rec, media, org, visit);
    }
  }
  if (!nested) hit = shape_hit<R, C>(sv.shape(obj_index), obj_index, local, t_min, t_max, rec, org.leaf() == obj_index ? org.side() : -2);
  if (!hit) return false;
  if constexpr (C::has(F_XFORM)) {
    record_to_world(ob, local.d, moved.d, rec);
    if (ob.flags & OBJ_FRONT_CHAIN) {  // a re-centred sphere: front_face as RotateY's mixed-frame call would have left it (src/hittable.rs:236-239)
      const R dlx = ob.cos_theta * ray.d.x - ob.sin_theta * ray.d.z, dlz = ob.sin_theta * ray.d.x + ob.cos_theta * ray.d.z;
      rec.front = (dlx * rec.n.x + ray.d.y * rec.n.y + dlz * rec.n.z) < (R)0;
    }
  }
  return true;
}

template <class R, class C, int LEVEL>
RT_DEV bool object_hit(const SceneView<R, C>& sv, int obj_index, const Ray<R>& ray, R t_min, R t_max, Rec<R>& rec, Stream& media, Origin org,
                       uint32_t visit) {
  const DObject<R> ob = sv.object(obj_index);
  if constexpr (C::has(F_MEDIUM)) {
    if (ob.flags & OBJ_MEDIUM) {  // src/constant_medium.rs:41-111
      const R inf = Lim<R>::inf();
      Rec<R> rec1, rec2;
      if (!boundary_hit<R, C, LEVEL>(sv, ob, obj_index, ray, -inf, inf, rec1, media, Origin(), visit)) return false;
      R t_after = rec1.t + (R)0.0001;
      if constexpr (Guard<R>::on) {
        // f32: beyond |t| = 1024 the spacing of floats exceeds 0.0001 and the sum rounds back to rec1.t (a fog boundary of
        // radius 5000 has t of that size); "strictly beyond the entry point" is what the reference's f64 sum means there.
        if (!(t_after > rec1.t)) t_after = nextafterf(rec1.t, Lim<float>::inf());
      }
      if (!boundary_hit<R, C, LEVEL>(sv, ob, obj_index, ray, t_after, inf, rec2, media, Origin(), visit)) return false;
      R t1 = rec1.t, t2 = rec2.t;
      if (t1 < t_min) t1 = t_min;
      if (t2 > t_max) t2 = t_max;
      if (t1 >= t2) return false;
      if (t1 < (R)0) t1 = (R)0;
      R ray_length = length(ray.d);
      R distance_inside_boundary = (t2 - t1) * ray_length;
      const DShape<R> ids = sv.shape(obj_index);
      R hit_distance = ob.neg_inv_density * gen_log01(media, medium_slot((uint32_t)ids.obj_id, visit), (R)0);
      if (hit_distance > distance_inside_boundary) return false;
      rec.t = t1 + hit_distance / ray_length;
      rec.p = ray.at(rec.t);
      rec.n = V3<R>((R)1, (R)0, (R)0);
      rec.front = 1;
      rec.mat = ob.medium_material;  // u, v untouched
      rec.obj = ids.obj_id;
      rec.sub = ids.sub_id;
      rec.side = -1;
      rec.leaf = -1;
      return true;
    }
  }
  return boundary_hit<R, C, LEVEL>(sv, ob, obj_index, ray, t_min, t_max, rec, media, org, visit);
}

template <class R, class C, int LEVEL>
RT_DEV bool leaf_hit(const SceneView<R, C>& sv, uint32_t leaf_word, const Ray<R>& ray, R t_min, R t_max, Rec<R>& rec, Stream& media, Origin org,
                     uint32_t visit = 0u) {
  int obj = (int)(leaf_word & LEAF_INDEX_MASK);
  if constexpr (C::has(F_XFORM) || C::has(F_MEDIUM) || C::has(F_NESTED)) {
    if (!(leaf_word & LEAF_SIMPLE)) return object_hit<R, C, LEVEL>(sv, obj, ray, t_min, t_max, rec, media, org, visit | visit_bit(leaf_word, LEVEL));
  }
  return shape_hit<R, C>(sv.shape(obj), obj, ray, t_min, t_max, rec, org.leaf() == obj ? org.side() : -2);
}

// BVHNode::hit (src/bvh.rs:108-119) over the threaded list, one traversal per lane.  The running `closest` is the
// t_max every node of the reference's recursion would receive: it starts as the caller's t_max and becomes rec.t
// after each accepted leaf, and nodes are visited in the reference's left-then-right order.
// Box test of the per-lane walk.  Double: src/aabb.rs:30-46 as written.  Float: hardware reciprocals, slab distances as
// (bound - origin) * 1/d (the difference first: folding the origin into a multiply-add loses the distance to rounding as soon as
// the origin is far from zero compared with the box), interval logic by min/max, which agrees with the reference's form except
// for rays lying exactly in a slab plane.
template <class R> struct LaneBoxTest;
// Double: the reciprocals are the IEEE quotients 1/d (the refined hardware estimate returns exactly those).  When all three are
// finite and non-zero - every ray of a render - no product 0 * inf can occur, the reference's "swap if 1/d < 0" is the smaller /
// larger of the two plane distances (an inverted box never reaches the device as such: see build_image), its three early outs
// `t_max <= t_min` are one comparison at the end (t_min only grows, t_max only shrinks), and its compare-and-keep updates are
// max / min: the same answer for a third fewer instructions.  A ray with a zero direction component takes src/aabb.rs:30-46 as
// written, NaNs included.
template <> struct LaneBoxTest<double> {
  V3<double> inv;
  bool plain;
  // The four-wide walk may test its child boxes in single precision (wide_keys below): the ray's origin and reciprocals are of
  // ordinary size (2^-40 .. 2^40: no product of the test can overflow or vanish) and the scene allows it.
  bool narrow;
  RT_DEV explicit LaneBoxTest(const Ray<double>& ray, bool allow_narrow = false)
      : inv(rcp_refined(ray.d.x), rcp_refined(ray.d.y), rcp_refined(ray.d.z)) {
    plain = __builtin_isfinite(inv.x) & __builtin_isfinite(inv.y) & __builtin_isfinite(inv.z) & (inv.x != 0.0) & (inv.y != 0.0) & (inv.z != 0.0);
    const double big = 0x1p40, small = 0x1p-40;
    const double ax = __builtin_fabs(inv.x), ay = __builtin_fabs(inv.y), az = __builtin_fabs(inv.z);
    narrow = allow_narrow & plain & (ax <= big) & (ay <= big) & (az <= big) & (ax >= small) & (ay >= small) & (az >= small) &
             (__builtin_fabs(ray.o.x) <= big) & (__builtin_fabs(ray.o.y) <= big) & (__builtin_fabs(ray.o.z) <= big);
  }
  // `exact`, `i`: DeviceScene::exact_boxes and the link's index - a box whose float form lets the ray in is tested again as the
  // caller gave it (scenes that have no such array: one scalar branch)
  RT_DEV bool hit(const BoxLink& f, const Ray<double>& ray, double t_min, double t_max, const double* exact = nullptr, int i = 0) const {
    DLink<double> b;  // the stored box, widened to double (exact)
    for (int k = 0; k < 3; ++k) { b.bmin[k] = (double)f.bmin[k]; b.bmax[k] = (double)f.bmax[k]; }
    bool h = test(b, ray, t_min, t_max);
    if (exact != nullptr) {
      if (h) {
        const double* e = exact + 6 * (size_t)i;
        for (int k = 0; k < 3; ++k) { b.bmin[k] = e[k]; b.bmax[k] = e[3 + k]; }
        h = test(b, ray, t_min, t_max);
      }
    }
    return h;
  }
  RT_DEV bool test(const DLink<double>& b, const Ray<double>& ray, double t_min, double t_max) const {
    if (plain) {
      const double x0 = (b.bmin[0] - ray.o.x) * inv.x, x1 = (b.bmax[0] - ray.o.x) * inv.x;
      const double y0 = (b.bmin[1] - ray.o.y) * inv.y, y1 = (b.bmax[1] - ray.o.y) * inv.y;
      const double z0 = (b.bmin[2] - ray.o.z) * inv.z, z1 = (b.bmax[2] - ray.o.z) * inv.z;
      const double tn = rt_max(rt_max(rt_min(x0, x1), rt_min(y0, y1)), rt_max(rt_min(z0, z1), t_min));
      const double tf = rt_min(rt_min(rt_max(x0, x1), rt_max(y0, y1)), rt_min(rt_max(z0, z1), t_max));
      return tn < tf;
    }
    return aabb_hit(b, ray, inv, t_min, t_max);
  }
};
template <> struct LaneBoxTest<float> {
  V3<float> inv;
  RT_DEV explicit LaneBoxTest(const Ray<float>& ray, bool = false)
      : inv(__builtin_amdgcn_rcpf(ray.d.x), __builtin_amdgcn_rcpf(ray.d.y), __builtin_amdgcn_rcpf(ray.d.z)) {}
  RT_DEV bool hit(const DLink<float>& b, const Ray<float>& ray, float t_min, float t_max, const double* = nullptr, int = 0) const {
    const float x0 = (b.bmin[0] - ray.o.x) * inv.x, x1 = (b.bmax[0] - ray.o.x) * inv.x;
    const float y0 = (b.bmin[1] - ray.o.y) * inv.y, y1 = (b.bmax[1] - ray.o.y) * inv.y;
    const float z0 = (b.bmin[2] - ray.o.z) * inv.z, z1 = (b.bmax[2] - ray.o.z) * inv.z;
    const float tn = rt_max(rt_max(rt_min(x0, x1), rt_min(y0, y1)), rt_max(rt_min(z0, z1), t_min));
    const float tf = rt_min(rt_min(rt_max(x0, x1), rt_max(y0, y1)), rt_min(rt_max(z0, z1), t_max));
    return tn < tf;
  }
};

template <class R, class C, int LEVEL>
RT_DEV bool traverse_lane(const SceneView<R, C>& sv, int start, const Ray<R>& ray, R t_min, R t_max, Rec<R>& rec, Stream& media, Origin org,
                          uint32_t visit, Visits* tally) {
  const LaneBoxTest<R> box(ray);
  R closest = t_max;
  bool any = false;
  int i = start;
  while (i >= 0) {
    BoxLink lk = sv.link(i);
    // Box entries first, until every lane of the wave stands on a leaf (or has left the tree): the leaf code below is then
    // entered by the lanes together instead of once per lane that happens to reach a leaf.
    while (!(lk.hit & LEAF_FLAG)) {
      if (tally) ++tally->boxes;
      i = box.hit(lk, ray, t_min, closest, sv.exact_boxes, i) ? (int)lk.hit : lk.miss;
      if (i < 0) break;
      lk = sv.link(i);
    }
    if (i < 0) break;
    if (tally) ++tally->leaves;
    if (leaf_hit<R, C, LEVEL>(sv, lk.hit, ray, t_min, closest, rec, media, org, visit)) { closest = rec.t; any = true; }
    i = lk.miss;
  }
  return any;
}

// ---- four-wide walk (WALK_LANE_WIDE) ----
// Closest hit over the library's own tree in its four-wide form.  A lane keeps the child it looks at next and a stack of the
// children still to look at (LDS, [entry][lane] so that the lanes of a wave never share a bank); a step on a node tests the four
// child boxes against [t_min, closest], goes on with the nearest child that passed and pushes the others.  A
// popped node is tested against the `closest` of that moment like any other, so what a nearer hit has made unreachable falls
// away one level down.  The closest hit does not depend on the order leaves are tested in (only which of two exactly coincident
// surfaces is reported does), which is why only the library's own tree, never the reference's, takes this walk.
RT_DEV void wide_stack_store(uint32_t* base, uint32_t entry, uint32_t lane, uint32_t child) {
  typedef __attribute__((address_space(3))) uint32_t* lds32;
  lds32 w = (lds32)(uintptr_t)(uint32_t)(uintptr_t)base;  // the low 32 bits of a generic LDS address are the LDS offset
  w[entry * 64u + lane] = child;
}
RT_DEV uint32_t wide_stack_load(const uint32_t* base, uint32_t entry, uint32_t lane) {
  typedef const __attribute__((address_space(3))) uint32_t* lds32;
  lds32 w = (lds32)(uintptr_t)(uint32_t)(uintptr_t)base;
  return w[entry * 64u + lane];
}

// Distance at which the ray enters child c's box, +infinity when it misses it inside [t_min, t_max] (an unused slot's box is the
// point at +infinity, which fails the test).  The distance only orders the children.
RT_DEV float wide_child_key(const LaneBoxTest<float>& box, const WideNode& nd, int c, const Ray<float>& ray, float t_min, float t_max) {
  const float x0 = (nd.lo[0][c] - ray.o.x) * box.inv.x, x1 = (nd.hi[0][c] - ray.o.x) * box.inv.x;
  const float y0 = (nd.lo[1][c] - ray.o.y) * box.inv.y, y1 = (nd.hi[1][c] - ray.o.y) * box.inv.y;
  const float z0 = (nd.lo[2][c] - ray.o.z) * box.inv.z, z1 = (nd.hi[2][c] - ray.o.z) * box.inv.z;
  const float tn = rt_max(rt_max(rt_min(x0, x1), rt_min(y0, y1)), rt_max(rt_min(z0, z1), t_min));
  const float tf = rt_min(rt_min(rt_max(x0, x1), rt_max(y0, y1)), rt_min(rt_max(z0, z1), t_max));
  return tn < tf ? tn : Lim<float>::inf();
}
RT_DEV float wide_child_key(const LaneBoxTest<double>& box, const WideNode& nd, int c, const Ray<double>& ray, double t_min, double t_max) {
  if (box.plain) {
    const double x0 = ((double)nd.lo[0][c] - ray.o.x) * box.inv.x, x1 = ((double)nd.hi[0][c] - ray.o.x) * box.inv.x;
    const double y0 = ((double)nd.lo[1][c] - ray.o.y) * box.inv.y, y1 = ((double)nd.hi[1][c] - ray.o.y) * box.inv.y;
    const double z0 = ((double)nd.lo[2][c] - ray.o.z) * box.inv.z, z1 = ((double)nd.hi[2][c] - ray.o.z) * box.inv.z;
    const double tn = rt_max(rt_max(rt_min(x0, x1), rt_min(y0, y1)), rt_max(rt_min(z0, z1), t_min));
    const double tf = rt_min(rt_min(rt_max(x0, x1), rt_max(y0, y1)), rt_min(rt_max(z0, z1), t_max));
    return tn < tf ? rt_min((float)tn, 3.0e38f) : Lim<float>::inf();  // finite: +infinity stands for a miss
  }
  BoxLink b;  // a ray with a zero direction component: the reference's test as written decides, the order is arbitrary
  for (int k = 0; k < 3; ++k) { b.bmin[k] = nd.lo[k][c]; b.bmax[k] = nd.hi[k][c]; }
  return box.hit(b, ray, t_min, t_max) ? 0.0f : Lim<float>::inf();
}

// The four child keys of a node.  Single precision: child by child.
RT_DEV void wide_keys(const LaneBoxTest<float>& box, const WideNode& nd, const Ray<float>& ray, float t_min, float t_max, float key[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) key[c] = wide_child_key(box, nd, c, ray, t_min, t_max);
}
// Double precision.  The boxes of the library's tree decide nothing but which leaves get tested, so their test may err on the
// side of entering a box - and then it can run in single precision, where an operation issues at half the cost of a double one
// and the node's floats need no conversion: three quarters of a node step's instructions were this test.  With u = 2^-24, the
// origin rounded to a float is off by at most u |o|, the reciprocal by u |1/d|, and the difference and the product round once
// each: a plane distance t computed this way differs from (c - o) / d by at most 3.01 u |t| + 1.01 u |o / d|.  Each axis' near
// distance is lowered and its far distance raised by e = 4 u |o / d|, and after the intervals of the three axes, t_min and the
// closest hit so far (rounded to a float: off by u) are combined, the lower end is lowered and the upper end raised by 4 u of
// themselves once more: whenever the double-precision test enters the box, so does this one.  An unused slot (a box at
// +infinity) ends in a NaN on one side and fails the comparison.  A ray whose numbers are not of ordinary size (`narrow`)
// takes the double-precision test.
RT_DEV void wide_keys(const LaneBoxTest<double>& box, const WideNode& nd, const Ray<double>& ray, double t_min, double t_max, float key[4]) {
  if (box.narrow) {
    const float ox = (float)ray.o.x, oy = (float)ray.o.y, oz = (float)ray.o.z;
    const float ix = (float)box.inv.x, iy = (float)box.inv.y, iz = (float)box.inv.z;
    const float ex = __builtin_fabsf(ox * ix) * 0x1p-22f, ey = __builtin_fabsf(oy * iy) * 0x1p-22f, ez = __builtin_fabsf(oz * iz) * 0x1p-22f;
    const float lo = (float)t_min, hi = (float)t_max;
    const float inf = Lim<float>::inf();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x0 = (nd.lo[0][c] - ox) * ix, x1 = (nd.hi[0][c] - ox) * ix;
      const float y0 = (nd.lo[1][c] - oy) * iy, y1 = (nd.hi[1][c] - oy) * iy;
      const float z0 = (nd.lo[2][c] - oz) * iz, z1 = (nd.hi[2][c] - oz) * iz;
      const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1) - ex, __builtin_fminf(y0, y1) - ey), __builtin_fmaxf(__builtin_fminf(z0, z1) - ez, lo));
      const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1) + ex, __builtin_fmaxf(y0, y1) + ey), __builtin_fminf(__builtin_fmaxf(z0, z1) + ez, hi));
      const float tn_low = __builtin_fmaf(-0x1p-22f, __builtin_fabsf(tn), tn), tf_high = __builtin_fmaf(0x1p-22f, __builtin_fabsf(tf), tf);
      key[c] = tn_low < tf_high ? tn : inf;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) key[c] = wide_child_key(box, nd, c, ray, t_min, t_max);
  }
}

template <class R> struct WideLane {
  uint32_t cur = kWideNone;  // node index, leaf word (LEAF_FLAG), or kWideNone: no walk in progress
  uint32_t sp = 0;
  RT_DEV bool idle() const { return cur == kWideNone; }
  RT_DEV bool at_box() const { return !(cur & LEAF_FLAG); }
  RT_DEV bool at_leaf() const { return (cur & LEAF_FLAG) != 0u && cur != kWideNone; }
  RT_DEV void start() { cur = 0u; sp = 0u; }
  RT_DEV void pop(const uint32_t* stack) {
    cur = kWideNone;
    if (sp > 0u) { --sp; cur = wide_stack_load(stack, sp, threadIdx.x & 63u); }
  }
  // one step on the node `cur`
  RT_DEV void box_step(const WideNode* wide, uint32_t* stack, const LaneBoxTest<R>& box, const Ray<R>& ray, R t_min, R closest) {
    const uint32_t lane = threadIdx.x & 63u;
#ifdef RT_WIDE_Q8
    WideNode nd;
    if constexpr (sizeof(R) == sizeof(float)) {
      const WideNodeQ q = reinterpret_cast<const WideNodeQ*>(wide)[cur];
      const float inf_ = Lim<float>::inf();
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        nd.child[c] = q.child[c];
        const bool used = q.child[c] != kWideNone;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float lo = q.org[a] + (float)((q.qlo[a] >> (8 * c)) & 255u) * q.cell[a];
          const float hi = q.org[a] + (float)((q.qhi[a] >> (8 * c)) & 255u) * q.cell[a];
          nd.lo[a][c] = used ? lo : inf_;
          nd.hi[a][c] = used ? hi : inf_;
        }
      }
    } else {
      nd = wide[cur];
    }
#else
    const WideNode nd = wide[cur];
#endif
    float key[4];
    uint32_t ch[4];
    wide_keys(box, nd, ray, t_min, closest, key);
#pragma unroll
    for (int c = 0; c < 4; ++c) ch[c] = nd.child[c];
    auto order = [&](int a, int b) {
      const bool s = key[b] < key[a];
      const float ka = s ? key[b] : key[a], kb = s ? key[a] : key[b];
      const uint32_t ca = s ? ch[b] : ch[a], cb = s ? ch[a] : ch[b];
      key[a] = ka; key[b] = kb; ch[a] = ca; ch[b] = cb;
    };
    // three exchanges bring the nearest child to slot 0; the others go on the stack as those exchanges left them (a full sort is
    // two more exchanges a step and measures 2 % slower: with more than two children hit their order seldom matters)
    order(0, 1); order(2, 3); order(0, 2);
    const float inf = Lim<float>::inf();
#pragma unroll
    for (int c = 3; c >= 1; --c) {
      if (key[c] < inf && sp < kWideStack) {  // the host guarantees room (kWideStack); the bound keeps a wave inside its own stack regardless
        wide_stack_store(stack, sp, lane, ch[c]);
        ++sp;
      }
    }
    if (key[0] < inf) cur = ch[0];
    else pop(stack);
  }
};

}  // namespace rtd
