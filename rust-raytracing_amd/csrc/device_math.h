// Device-side scalar / vector arithmetic, templated on the arithmetic type R (float or double),
// and the counter-based random streams.  Operation order follows the reference's f64 code
// (src/vec3.rs) so that the R=double build tracks a CPU f64 evaluation to rounding error.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtd {

#define RT_DEV __device__ __forceinline__
#define RT_HD __host__ __device__ __forceinline__

template <class R> struct Lim;
template <> struct Lim<float> {
  RT_DEV static float inf() { return __builtin_inff(); }
};
template <> struct Lim<double> {
  RT_DEV static double inf() { return __builtin_inf(); }
};

// Elementary functions.  The float versions use the hardware approximations (v_rcp/v_sqrt/v_sin/v_log,
// about 1 ulp) where the reference calls an f64 libm routine; double versions are the IEEE/ocml ones.
RT_DEV float rt_sqrt(float x) { return __builtin_sqrtf(x); }
RT_DEV double rt_sqrt(double x) { return __builtin_sqrt(x); }
RT_DEV float rt_abs(float x) { return __builtin_fabsf(x); }
RT_DEV double rt_abs(double x) { return __builtin_fabs(x); }
RT_DEV float rt_floor(float x) { return __builtin_floorf(x); }
RT_DEV double rt_floor(double x) { return __builtin_floor(x); }
RT_DEV float rt_min(float a, float b) { return __builtin_fminf(a, b); }
RT_DEV double rt_min(double a, double b) { return __builtin_fmin(a, b); }
RT_DEV float rt_max(float a, float b) { return __builtin_fmaxf(a, b); }
RT_DEV double rt_max(double a, double b) { return __builtin_fmax(a, b); }
RT_DEV float rt_sin(float x) { return sinf(x); }
RT_DEV double rt_sin(double x) { return sin(x); }
RT_DEV float rt_log(float x) { return logf(x); }
RT_DEV double rt_log(double x) { return log(x); }
RT_DEV float rt_acos(float x) { return acosf(x); }
RT_DEV double rt_acos(double x) { return acos(x); }
RT_DEV float rt_atan2(float y, float x) { return atan2f(y, x); }
RT_DEV double rt_atan2(double y, double x) { return atan2(y, x); }

template <class R> struct V3 {
  R x, y, z;
  RT_HD V3() : x(0), y(0), z(0) {}
  RT_HD V3(R a, R b, R c) : x(a), y(b), z(c) {}
};
template <class R> RT_DEV V3<R> operator+(V3<R> a, V3<R> b) { return V3<R>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class R> RT_DEV V3<R> operator-(V3<R> a, V3<R> b) { return V3<R>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class R> RT_DEV V3<R> operator*(V3<R> a, V3<R> b) { return V3<R>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <class R> RT_DEV V3<R> operator*(R f, V3<R> v) { return V3<R>(v.x * f, v.y * f, v.z * f); }  // src/vec3.rs:200-206
template <class R> RT_DEV V3<R> operator-(V3<R> v) { return V3<R>(-v.x, -v.y, -v.z); }
template <class R> RT_DEV R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // src/vec3.rs:58-60
template <class R> RT_DEV R length_squared(V3<R> v) { return v.x * v.x + v.y * v.y + v.z * v.z; }         // src/vec3.rs:49-51
template <class R> RT_DEV R length(V3<R> v) { return rt_sqrt(length_squared(v)); }                        // src/vec3.rs:45-47
template <class R> RT_DEV V3<R> vdiv(V3<R> v, R f) { return ((R)1 / f) * v; }                             // src/vec3.rs:224-230
template <class R> RT_DEV V3<R> unit_vector(V3<R> v) { return vdiv(v, length(v)); }                       // src/vec3.rs:77-79
template <class R> RT_DEV bool near_zero(V3<R> v) {                                                       // src/vec3.rs:53-56
  const R s = (R)1e-8;
  return rt_abs(v.x) < s && rt_abs(v.y) < s && rt_abs(v.z) < s;
}
template <class R> RT_DEV V3<R> reflect(V3<R> v, V3<R> n) { return v - ((R)2 * dot(v, n)) * n; }           // src/vec3.rs:113-115
template <class R> RT_DEV V3<R> refract(V3<R> uv, V3<R> n, R etai_over_etat) {                            // src/vec3.rs:117-122
  R cos_theta = rt_min(dot(-uv, n), (R)1);
  V3<R> r_out_perp = etai_over_etat * (uv + cos_theta * n);
  V3<R> r_out_parallel = (-rt_sqrt(rt_abs((R)1 - length_squared(r_out_perp)))) * n;
  return r_out_perp + r_out_parallel;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).
// Stands in for rand::thread_rng() (src/render.rs:226, src/camera.rs:69, src/vec3.rs:20,25,104,
// src/constant_medium.rs:50, src/material.rs:127).
// ---------------------------------------------------------------------------------------------
RT_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t& o0, uint32_t& o1,
                          uint32_t& o2, uint32_t& o3) {
#pragma unroll
#ifndef RT_X_ROUNDS
#define RT_X_ROUNDS 10
#endif
  for (int r = 0; r < RT_X_ROUNDS; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

enum : uint32_t { STAGE_CAMERA = 0, STAGE_MEDIA = 1, STAGE_SCATTER = 2 };

// One stream per (pixel, sample, bounce, stage): key = (pixel, sample), counter = (block, bounce | stage << 16,
// seed_lo, seed_hi).  Draw i is the 64-bit word (i & 1) of block i / 2 (word k = out[2k+1] << 32 | out[2k]).
struct Stream {
  uint32_t k0, k1, c1, c2, c3, idx, w_lo, w_hi;
  RT_DEV Stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t stage)
      : k0(pixel), k1(sample), c1(bounce | (stage << 16)), c2((uint32_t)seed), c3((uint32_t)(seed >> 32)), idx(0), w_lo(0), w_hi(0) {}
  RT_DEV uint64_t next_u64() {
    uint64_t w;
    if ((idx & 1u) == 0) {
      uint32_t o0, o1, o2, o3;
      philox4x32_10(idx >> 1, c1, c2, c3, k0, k1, o0, o1, o2, o3);
      w_lo = o2; w_hi = o3;
      w = ((uint64_t)o1 << 32) | o0;
    } else {
      w = ((uint64_t)w_hi << 32) | w_lo;
    }
    ++idx;
    return w;
  }
};

// rand 0.8.5 value conversions: Standard f64 keeps 53 bits; gen_range builds a 52-bit mantissa in [1,2) - 1.
// The float build takes the top 24 / 23 bits of the same word, so both builds consume the stream alike.
template <class R> struct Uni;
template <> struct Uni<double> {
  RT_DEV static double u01(uint64_t w) { return (double)(w >> 11) * 0x1.0p-53; }
  RT_DEV static double m01(uint64_t w) { return (double)(w >> 12) * 0x1.0p-52; }
};
template <> struct Uni<float> {
  RT_DEV static float u01(uint64_t w) { return (float)(uint32_t)(w >> 40) * 0x1.0p-24f; }
  RT_DEV static float m01(uint64_t w) { return (float)(uint32_t)(w >> 41) * 0x1.0p-23f; }
};
RT_DEV float next_toward_zero(float x) { return __uint_as_float(__float_as_uint(x) - 1u); }
RT_DEV double next_toward_zero(double x) { return __longlong_as_double(__double_as_longlong(x) - 1ll); }

template <class R> RT_DEV R gen01(Stream& s) { return Uni<R>::u01(s.next_u64()); }
// ln(U) for the next 53-bit uniform U (free-path sampling, src/constant_medium.rs:91).  A float cannot represent U close
// to 1, where ln(U) ~ -(1 - U) is tiny, so the float build forms 1 - U exactly from the integer and uses log1p there.
RT_DEV double gen_log01(Stream& s, double) { return rt_log(Uni<double>::u01(s.next_u64())); }
RT_DEV float gen_log01(Stream& s, float) {
  uint64_t k = s.next_u64() >> 11;
  if (k >= (1ull << 52)) return log1pf(-((float)((1ull << 53) - k) * 0x1.0p-53f));
  return rt_log((float)k * 0x1.0p-53f);
}
template <class R> RT_DEV R gen_range(Stream& s, R lo, R hi) {  // half-open [lo, hi)
  R scale = hi - lo;
  for (;;) {
    R res = Uni<R>::m01(s.next_u64()) * scale + lo;
    if (res < hi) return res;
    scale = next_toward_zero(scale);
  }
}
template <class R> RT_DEV V3<R> random_range_vec(Stream& s, R lo, R hi) {  // src/vec3.rs:24-31
  R a = gen_range<R>(s, lo, hi);
  R b = gen_range<R>(s, lo, hi);
  R c = gen_range<R>(s, lo, hi);
  return V3<R>(a, b, c);
}
template <class R> RT_DEV V3<R> random_in_unit_sphere(Stream& s) {  // src/vec3.rs:81-88
#ifdef RT_X_BOUND
  for (int guard = 0; guard < 64; ++guard) {
#else
  for (;;) {
#endif
    V3<R> p = random_range_vec<R>(s, (R)-1, (R)1);
#ifdef RT_X_NOREJECT
    return p;
#endif
    if (length_squared(p) < (R)1) return p;
  }
  return V3<R>((R)0.5, (R)0, (R)0);
}
template <class R> RT_DEV V3<R> random_unit_vector(Stream& s) { return unit_vector(random_in_unit_sphere<R>(s)); }  // src/vec3.rs:90-92
template <class R> RT_DEV V3<R> random_in_unit_disk(Stream& s) {  // src/vec3.rs:103-111
  for (;;) {
    R a = gen_range<R>(s, (R)-1, (R)1);
    R b = gen_range<R>(s, (R)-1, (R)1);
    V3<R> p(a, b, (R)0);
    if (length_squared(p) < (R)1) return p;
  }
}

}  // namespace rtd
