// Shading for one hit per lane: Material::emitted / Material::scatter (src/material.rs), Texture::value
// (src/texture.rs), Perlin noise (src/perlin.rs), the miss shader (src/color.rs:41-51), primary-ray
// generation (src/render.rs:230-234 + src/camera.rs:62-72) and the final colour conversion (src/color.rs:14-27).
#pragma once
#include "device_trace_deferred.h"

namespace rtd {

template <class R> RT_DEV R perlin_noise(const DPerlin<R>& pn, V3<R> pt) {  // src/perlin.rs:32-55,116-139
  R fx = rt_floor(pt.x), fy = rt_floor(pt.y), fz = rt_floor(pt.z);
  R u = pt.x - fx, v = pt.y - fy, w = pt.z - fz;
  int i = (int)fx, j = (int)fy, k = (int)fz;
  R uu = u * u * ((R)3 - (R)2 * u);
  R vv = v * v * ((R)3 - (R)2 * v);
  R ww = w * w * ((R)3 - (R)2 * w);
  R accum = 0;
#pragma unroll
  for (int di = 0; di < 2; ++di)
#pragma unroll
    for (int dj = 0; dj < 2; ++dj)
#pragma unroll
      for (int dk = 0; dk < 2; ++dk) {
        int idx = pn.perm[0][(i + di) & 255] ^ pn.perm[1][(j + dj) & 255] ^ pn.perm[2][(k + dk) & 255];
        V3<R> c(pn.ran[idx][0], pn.ran[idx][1], pn.ran[idx][2]);
        R fi = (R)di, fj = (R)dj, fk = (R)dk;
        V3<R> weight_v(u - fi, v - fj, w - fk);
        accum += (fi * uu + ((R)1 - fi) * ((R)1 - uu)) * (fj * vv + ((R)1 - fj) * ((R)1 - vv)) *
                 (fk * ww + ((R)1 - fk) * ((R)1 - ww)) * dot(c, weight_v);
      }
  return accum;
}

template <class R> RT_DEV R perlin_turbulence(const DPerlin<R>& pn, V3<R> pt, int depth) {  // src/perlin.rs:60-72
  R accum = 0;
  V3<R> temp_p = pt;
  R weight = 1;
  for (int i = 0; i < depth; ++i) {
    accum += weight * perlin_noise(pn, temp_p);
    weight *= (R)0.5;
    temp_p = (R)2 * temp_p;
  }
  return rt_abs(accum);
}

RT_DEV uint32_t sat_u32(float x) { return !(x > 0.0f) ? 0u : (x >= 4294967295.0f ? 4294967295u : (uint32_t)x); }
RT_DEV uint32_t sat_u32(double x) { return !(x > 0.0) ? 0u : (x >= 4294967295.0 ? 4294967295u : (uint32_t)x); }

// Texture::value.  LEVEL 0 may be a checker whose two sub-textures are evaluated at LEVEL 1.  Only an image reads (u, v): a
// sphere's are worked out here (get_sphere_uv, an arc cosine and an arc tangent), for the lanes that look a texel up, and not
// for every sphere a wave has hit.
template <class R, class C, int LEVEL> RT_DEV V3<R> texture_value(const SceneView<R, C>& sv, int tex, const Rec<R>& rec) {
  const DTexture<R> t = sv.textures[tex];
  const V3<R> p = rec.p;
  switch (t.tag) {
    case TEX_SOLID: return V3<R>(t.color[0], t.color[1], t.color[2]);  // src/texture.rs:29-33
    case TEX_CHECKER: {                                               // src/texture.rs:54-63
      if constexpr (LEVEL == 0 && C::has(F_CHECKER)) {
        // sines < 0: an odd number of negative factors and none of them zero (sine_negative); arguments too large for that -
        // coordinates beyond 10^5 - take the product of the library's sines
        bool zero = false, big = false;
        bool negative = sine_negative((R)10 * p.x, zero, big) != sine_negative((R)10 * p.y, zero, big);
        negative = (negative != sine_negative((R)10 * p.z, zero, big)) && !zero;
        if (__ballot(big) != 0ull) {
          if (big) {
            R sines = (R)1;
#pragma unroll 1
            for (int k = 0; k < 3; ++k) sines = sines * rt_sin((R)10 * (k == 0 ? p.x : k == 1 ? p.y : p.z));
            negative = sines < (R)0;
          }
        }
        return texture_value<R, C, 1>(sv, negative ? t.odd : t.even, rec);
      } else {
        return V3<R>((R)0, (R)0, (R)0);
      }
    }
    case TEX_NOISE: if constexpr (C::has(F_NOISE)) {  // src/texture.rs:79-85
      R s = (R)1 + rt_sin(t.scale * p.z + (R)10 * perlin_turbulence(sv.perlins[t.perlin], p, 7));
      return (V3<R>((R)1, (R)1, (R)1) * V3<R>((R)0.5, (R)0.5, (R)0.5)) * V3<R>(s, s, s);
    } break;
    case TEX_IMAGE: if constexpr (C::has(F_IMAGE)) {  // src/texture.rs:128-159
      if (t.image < 0) return V3<R>((R)0, (R)1, (R)1);
      const DImage im = sv.images[t.image];
      Rec<R> final_rec = rec;
      final_rec.finish_uv();
      R uc = rt_min(rt_max(final_rec.u, (R)0), (R)1);
      R vc = (R)1 - rt_min(rt_max(final_rec.v, (R)0), (R)1);
      uint32_t i = sat_u32(uc * (R)im.width);
      uint32_t j = sat_u32(vc * (R)im.height);
      if (i >= (uint32_t)im.width) i = (uint32_t)im.width - 1u;
      if (j >= (uint32_t)im.height) j = (uint32_t)im.height - 1u;
      const uint8_t* px = im.rgb + ((size_t)j * (size_t)im.width + i) * 3u;
      R color_scale = (R)1 / (R)255;
      return V3<R>(color_scale * (R)px[0], color_scale * (R)px[1], color_scale * (R)px[2]);
    } break;
    default: break;
  }
  return V3<R>((R)0, (R)0, (R)0);
}

template <class R> RT_DEV R schlick_reflectance(R cosine, R ref_idx) {  // src/material.rs:104-109, powi(5) = x * (x^2)^2
  R r0 = ((R)1 - ref_idx) / ((R)1 + ref_idx);
  r0 = r0 * r0;
  R x = (R)1 - cosine;
  R x2 = x * x;
  R x4 = x2 * x2;
  return r0 + ((R)1 - r0) * (x * x4);
}

// Colour of a material whose `texture` field is a texture index, or ~index when that texture is a solid colour, which the
// material then carries in `albedo` (scenes without any other kind of texture never look at the texture records).
template <class R, class C> RT_DEV V3<R> material_color(const SceneView<R, C>& sv, const DMaterial<R>& m, const Rec<R>& rec) {
  if (!C::has(F_CHECKER | F_NOISE | F_IMAGE) || 0 > m.texture) return V3<R>(m.albedo[0], m.albedo[1], m.albedo[2]);
  return texture_value<R, C, 0>(sv, m.texture, rec);
}

// material.emitted + material.scatter for a final hit record (src/render.rs:162-176).
// Returns true when a scattered ray exists; `emitted` is always written.
template <class R, class C, class S>
RT_DEV bool shade_hit(const SceneView<R, C>& sv, const Ray<R>& ray_in, Rec<R>& rec, const S& rng, V3<R>& emitted, V3<R>& attenuation,
                      Ray<R>& scattered) {
  const DMaterial<R> m = sv.materials[rec.mat];
  emitted = V3<R>((R)0, (R)0, (R)0);
  scattered.o = rec.p;
  scattered.time = ray_in.time;
  switch (m.tag) {
    case MAT_LAMBERTIAN: {  // src/material.rs:43-57
      V3<R> dir = rec.n + random_unit_vector<R, C::relaxed()>(rng);
      // practically never true: a branch the whole wave skips instead of three selects in every lane
      const bool degenerate = near_zero(dir);
      if (__ballot(degenerate) != 0ull) {
        if (degenerate) dir = rec.n;
      }
      scattered.d = dir;
      attenuation = material_color<R, C>(sv, m, rec);
      return true;
    }
    case MAT_METAL: if constexpr (C::has(F_METAL)) {  // src/material.rs:75-90
      V3<R> reflected = reflect(unit_vector(ray_in.d), rec.n);
      scattered.d = reflected + m.fuzz * random_in_unit_sphere<R>(rng);
      attenuation = V3<R>(m.albedo[0], m.albedo[1], m.albedo[2]);
      return dot(scattered.d, rec.n) > (R)0;
    } break;
    case MAT_DIELECTRIC: if constexpr (C::has(F_DIELECTRIC)) {  // src/material.rs:113-139
      attenuation = V3<R>((R)1, (R)1, (R)1);
      R refraction_ratio = rec.front ? (R)1 / m.ir : m.ir;
      V3<R> unit_direction = unit_vector(ray_in.d);
      R cos_theta = rt_min(dot(-unit_direction, rec.n), (R)1);
      R sin_theta = rt_sqrt((R)1 - cos_theta * cos_theta);
      bool cannot_refract = refraction_ratio * sin_theta > (R)1;
      bool do_reflect = cannot_refract;
      if (!do_reflect) do_reflect = schlick_reflectance(cos_theta, refraction_ratio) > Block<R>(rng, 0u).u01(0);  // `||` short-circuit
      scattered.d = do_reflect ? reflect(unit_direction, rec.n) : refract(unit_direction, rec.n, refraction_ratio);
      return true;
    } break;
    case MAT_LIGHT: {  // src/material.rs:159-167
      emitted = material_color<R, C>(sv, m, rec);
      return false;
    }
    case MAT_ISOTROPIC: if constexpr (C::has(F_MEDIUM)) {  // src/material.rs:186-195
      scattered.d = random_in_unit_sphere<R>(rng);
      attenuation = material_color<R, C>(sv, m, rec);
      return true;
    } break;
    default: break;
  }
  return false;
}

template <class R, class C> RT_DEV V3<R> background_color(const DBackground<R>& bg, V3<R> direction) {  // src/color.rs:42-50
  if (bg.kind == 0 || !C::has(F_GRADIENT)) return bg.c0;
  V3<R> ud = unit_vector(direction);
  R t = (R)0.5 * (ud.y + (R)1);
  return ((R)1 - t) * bg.c1 + t * bg.c0;
}

// Row and column of a pixel index without an integer division: a float reciprocal estimate corrected by one step is exact
// while the index is exactly representable (images below 2^24 pixels); larger images divide.
RT_DEV void pixel_row_col(uint32_t pixel, uint32_t W, uint32_t& y, uint32_t& i) {
  if (pixel < (1u << 24)) {
    uint32_t q = (uint32_t)((float)pixel * __builtin_amdgcn_rcpf((float)W));
    int32_t r = (int32_t)(pixel - q * W);
    if (r < 0) { q -= 1u; r += (int32_t)W; }
    if (r >= (int32_t)W) { q += 1u; r -= (int32_t)W; }
    y = q; i = (uint32_t)r;
  } else {
    y = pixel / W; i = pixel - y * W;
  }
}
// (i + u) / (W - 1), src/render.rs:232-233.  Single precision multiplies by the reciprocal (wave-uniform, one rounding more).
// RELAXED: double precision multiplies by the (wave-uniform) reciprocal as well.
template <bool RELAXED> RT_DEV double jitter_coord(uint32_t i, double u, uint32_t wm1) {
  if constexpr (RELAXED) return ((double)i + u) * (1.0 / (double)wm1);
  else return ((double)i + u) / (double)wm1;
}
template <bool RELAXED> RT_DEV float jitter_coord(uint32_t i, float u, uint32_t wm1) { return ((float)i + u) * (1.0f / (float)wm1); }

// Primary ray of (pixel, sample): src/render.rs:230-234 then Camera::get_ray (src/camera.rs:62-72).
// pixel = y * W + x with y counted from the top row; the reference's j is H - 1 - y.
// `rng` is the camera stream of (pixel, sample).
// `y`, `i`: the pixel's row (from the top) and column.
template <class R, bool RELAXED, class S>
RT_DEV Ray<R> primary_ray_at(const DCamera<R>& cam, uint32_t W, uint32_t H, uint32_t y, uint32_t i, const S& rng) {
  uint32_t j = H - 1u - y;
  const Block<R> jitter(rng, 0u);
  R u_rand = jitter.u01(0);
  R v_rand = jitter.u01(1);
  R s = jitter_coord<RELAXED>(i, u_rand, W - 1u);
  R t = jitter_coord<RELAXED>(j, v_rand, H - 1u);
  // a pinhole camera (lens_radius == 0, the same in every lane) multiplies the disk sample by zero: the offset is a zero vector,
  // x - 0 = x and x + 0 = x, so neither the sample nor the offset is formed; draws are addressed, so skipping them moves nothing else
  Ray<R> ray;
  ray.d = cam.lower_left_corner + s * cam.horizontal + t * cam.vertical - cam.origin;
  ray.o = cam.origin;
  if (cam.lens_radius != (R)0) {
    const V3<R> rd = cam.lens_radius * random_in_unit_disk<R>(rng);
    const V3<R> offset = rd.x * cam.u + rd.y * cam.v;
    ray.d = ray.d - offset;
    ray.o = cam.origin + offset;
  }
  ray.time = jitter.range(2, cam.t_open, cam.t_close);
  return ray;
}
template <class R, bool RELAXED, class S>
RT_DEV Ray<R> primary_ray(const DCamera<R>& cam, uint32_t W, uint32_t H, uint32_t pixel, const S& rng) {
  uint32_t y, i;
  pixel_row_col(pixel, W, y, i);
  return primary_ray_at<R, RELAXED>(cam, W, H, y, i, rng);
}
template <class R, bool RELAXED = false>
RT_DEV Ray<R> primary_ray(const DCamera<R>& cam, uint32_t W, uint32_t H, uint32_t pixel, uint32_t sample, uint64_t seed) {
  return primary_ray<R, RELAXED>(cam, W, H, pixel, Stream(seed, pixel, sample, 0u, STAGE_CAMERA));
}

// color_to_rgb (src/color.rs:14-27); Rust `as u8` saturates and maps NaN to 0.
template <class R> RT_DEV uint8_t to_u8(R sum, R scale) {
  R x = rt_sqrt(scale * sum);
  if (x < (R)0) x = (R)0;
  if (x > (R)0.999) x = (R)0.999;
  R y = (R)256 * x;
  if (!(y > (R)0)) return 0;
  if (y >= (R)255) return 255;
  return (uint8_t)y;
}

}  // namespace rtd
