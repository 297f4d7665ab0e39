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
// min / max of values that are known not to be signalling NaNs (products and sums the hardware has already made canonical): the
// plain instruction, without the canonicalising self-maximum the compiler puts in front of fmin / fmax when it cannot tell.
RT_DEV double min_plain(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_DEV double max_plain(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_DEV float min_plain(float a, float b) { return __builtin_fminf(a, b); }
RT_DEV float max_plain(float a, float b) { return __builtin_fmaxf(a, b); }
RT_DEV float rt_sin(float x) { return sinf(x); }
RT_DEV double rt_sin(double x) { return sin(x); }
// sin(x) < 0 without the sine, for the checker texture (src/texture.rs:54-63), which looks at nothing but the sign of a product of
// three sines.  sin(t) = (-1)^k sin(r) with k = nearest integer to t / pi and r = t - k pi in [-pi/2, pi/2], where sin has the sign
// of r.  Double precision: pi is taken in three pieces of 33 bits, so that k * piece is exact for |k| < 2^20 and r comes out with an
// absolute error near 1e-25 - no double below 1e6 in magnitude lies that close to a multiple of pi, so the sign is the sign of the
// true sine; larger or non-finite arguments report `big` and the caller evaluates the library's sine.  `zero`: sin(t) == 0 (t == 0).
RT_DEV bool sine_negative(double t, bool& zero, bool& big) {
  big = big | !(__builtin_fabs(t) < 1.0e6);
  zero = zero | (t == 0.0);
  const double k = __builtin_rint(t * 0.318309886183790671538);
  double r = t - k * (2.0 * 1.57079632673412561417e+00);
  r = r - k * (2.0 * 6.07710050630396597660e-11);
  r = r - k * (2.0 * 2.02226624871116645580e-21);
  return (r < 0.0) != (((int)k & 1) != 0);
}
// Single precision (tolerance mode): the parity of floor(t / pi); within a float's rounding of a multiple of pi either answer is as
// good as the position it is asked for.
RT_DEV bool sine_negative(float t, bool& zero, bool&) {
  zero = zero | (t == 0.0f);
  return ((int)__builtin_floorf(t * 0.318309886f) & 1) != 0;
}
RT_DEV float rt_log(float x) { return logf(x); }
RT_DEV double rt_log(double x) { return log(x); }
RT_DEV float rt_acos(float x) { return acosf(x); }
RT_DEV double rt_acos(double x) { return acos(x); }
RT_DEV float rt_atan2(float y, float x) { return atan2f(y, x); }
RT_DEV double rt_atan2(double y, double x) { return atan2(y, x); }

// 1/x in double precision from the hardware estimate (about 26 bits) and two Newton steps: on gfx950 the result is the IEEE
// quotient 1.0/x for every finite non-zero x tried (2^20 random arguments, tools/rcp_probe.hip), at half the instructions of the
// division sequence.  Zero, infinite and NaN arguments, for which the Newton steps make NaN, get the division's answer (inf / 0 /
// NaN) from the instruction the division sequence itself ends with; it passes every other quotient through unchanged.
RT_DEV double rcp_refined(double x) {
  const double r0 = __builtin_amdgcn_rcp(x);
  const double r1 = __builtin_fma(__builtin_fma(-x, r0, 1.0), r0, r0);
  const double r2 = __builtin_fma(__builtin_fma(-x, r1, 1.0), r1, r1);
  return __builtin_amdgcn_div_fixup(r2, x, 1.0);
}
// The same with ONE Newton step: 0.6 ulp from the quotient on average, 11 at most (same probe), for the places that already multiply
// by a reciprocal where the reference divides and so differ from it in the last digit either way (the flat walk's plane distances).
RT_DEV double rcp_one_step(double x) {
  const double r0 = __builtin_amdgcn_rcp(x);
  const double r1 = __builtin_fma(__builtin_fma(-x, r0, 1.0), r0, r0);
  return __builtin_amdgcn_div_fixup(r1, x, 1.0);
}
// 1/sqrt(x), hardware estimate and two Newton steps: within 2 ulp (same probe).
RT_DEV double rsq_refined(double x) {
  const double s0 = __builtin_amdgcn_rsq(x);
  const double s1 = __builtin_fma(s0, __builtin_fma(-x * s0, 0.5 * s0, 0.5), s0);
  return __builtin_fma(s1, __builtin_fma(-x * s1, 0.5 * s1, 0.5), s1);
}

// One step: 1 ulp on average, 19 at most (same probe).  For the unit vector of a Lambertian bounce in the small-scene presets, whose direction
// then differs from the reference's sqrt-then-divide by parts in 10^16 - the next hit point by as much, and nothing downstream
// amplifies it there.
RT_DEV double rsq_one_step(double x) {
  const double s0 = __builtin_amdgcn_rsq(x);
  return __builtin_fma(s0, __builtin_fma(-x * s0, 0.5 * s0, 0.5), s0);
}

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
RT_DEV V3<double> vdiv(V3<double> v, double f) { return rcp_refined(f) * v; }  // (1.0 / f) * v of src/vec3.rs:224-230; the refined reciprocal is that quotient bit for bit
RT_DEV V3<float> vdiv(V3<float> v, float f) { return __builtin_amdgcn_rcpf(f) * v; }                      // hardware reciprocal (1 ulp)
RT_DEV V3<double> unit_vector(V3<double> v) { return vdiv(v, length(v)); }                                // src/vec3.rs:77-79
RT_DEV V3<float> unit_vector(V3<float> v) { return __builtin_amdgcn_rsqf(length_squared(v)) * v; }        // hardware 1/sqrt (1 ulp)
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
// Philox4x32-7 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11): the 4x32 round function and
// constants with seven rounds, the smallest count the paper reports as passing BigCrush (ten is its default, with extra margin).
// A block costs two 32x32->64 multiplies per round at a quarter of the plain issue rate, and random draws are a third of this
// kernel's instructions, so the three rounds are worth about a tenth of the frame time.
// Stands in for rand::thread_rng() (src/render.rs:226, src/camera.rs:69, src/vec3.rs:20,25,104,
// src/constant_medium.rs:50, src/material.rs:127).
// ---------------------------------------------------------------------------------------------
constexpr int kPhiloxRounds = 7;
// a ^ b ^ c in one instruction (gfx950's three-input bitwise operation with the truth table of a three-way exclusive or): a Philox
// round ends in two of them, and written as two two-input operations each they are a sixth of a block's instructions
RT_DEV uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
RT_DEV void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t& o0, uint32_t& o1,
                          uint32_t& o2, uint32_t& o3) {
#pragma unroll
  for (int r = 0; r < kPhiloxRounds; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
    uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
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

// The bits of a draw below the 32 (or 21) that Philox supplies come from a multiplicative hash of the Philox words themselves:
// they place a value inside an interval of 2^-32 (2^-21 for a unit-ball coordinate) of its range, where all that matters is
// that they are spread evenly, and a second Philox block per draw costs a tenth of the double-precision frame time.
RT_DEV uint32_t low_mix(uint32_t a, uint32_t b) {
  uint32_t m = (a ^ ((b << 16) | (b >> 16))) * 0x7FEB352Du;
  return m ^ (m >> 15);
}

// One stream per (pixel, sample, bounce, stage): key = (pixel, sample), counter = (block, bounce | stage << 16 | half << 20,
// seed_lo, seed_hi).  Draws are addressed, not consumed in sequence: draw slot s is the 64-bit word H[s & 3] << 32 | L[s & 3],
// H = Philox block s >> 2, L[j] = low_mix(H[j], H[(j + 1) & 3]) * 0x846CA68B.  Slots by stage:
//   CAMERA   0 = pixel jitter u, 1 = jitter v, 2 = shutter time; lens-disk candidate c = slots 4(1+c), 4(1+c)+1
//   SCATTER  unit-sphere candidate c = two words (w0, w1) of H block c >> 1 (words 0,1 for even c, 2,3 for odd c) cut into three
//            21-bit fields + the top 31 bits of the three words low3(w0, w1) (52-bit mantissas); the dielectric's uniform = slot 0
//   MEDIA    the free-path draw of ConstantMedium object k = slot k (independent of the order the tree visits objects in)
// A rejection loop reads "the first candidate c = 0, 1, 2, ... that is accepted" (src/vec3.rs:81-88,103-111).
// The float build uses the top 24 / 23 bits of a word, so it needs H only and it sees the same numbers, rounded, as the double build.
struct Stream {
#ifdef RT_STAMPS
  unsigned long long* stamp_acc = nullptr;  // diagnostic build: the kernel's phase accumulators, reachable from the hit routines
  unsigned long long* stamp_t = nullptr;
#endif
  uint32_t k0, k1, c1, c2, c3;
  RT_DEV Stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t stage)
      : k0(pixel), k1(sample), c1(bounce | (stage << 16)), c2((uint32_t)seed), c3((uint32_t)(seed >> 32)) {}
  RT_DEV void hi(uint32_t block, uint32_t h[4]) const { philox4x32(block, c1, c2, c3, k0, k1, h[0], h[1], h[2], h[3]); }
  // Low halves of the four draw words of a block whose high halves are h (see low_mix).
  RT_DEV static void lo_of(const uint32_t h[4], uint32_t l[4]) {
    l[0] = low_mix(h[0], h[1]) * 0x846CA68Bu; l[1] = low_mix(h[1], h[2]) * 0x846CA68Bu;
    l[2] = low_mix(h[2], h[3]) * 0x846CA68Bu; l[3] = low_mix(h[3], h[0]) * 0x846CA68Bu;
  }
  // where a rejection loop starts reading candidate blocks, and whether it has to stop before block b (never, for a plain stream)
  RT_DEV uint32_t first_block() const { return 0u; }
  RT_DEV bool out_of_turn(uint32_t) const { return false; }
};
// A stream whose H block 0 is already computed.  The path kernels evaluate that block once per turn for all lanes - a scatter
// stream in the lanes that continue a path, a camera stream in the lanes that start one - so both uses share one pass of Philox.
//
// A rejection loop on such a stream looks at `per_turn` blocks and then gives up for this turn, leaving in `resume` the block
// to go on from; the kernel keeps the lane's hit and calls again next turn with b0 = resume.  A wave repeats the loop until its
// unluckiest lane is done (five blocks on average for a full wave, when a lane needs 1.3), and blocks are addressed, so putting
// off the few lanes that are still looking after two blocks changes no result and saves most of those passes.
struct PrimedStream : Stream {
  uint32_t b0, per_turn;
  uint32_t h0[4];
  mutable uint32_t resume = 0u;  // != 0: the draw was not finished, continue at this block
  RT_DEV PrimedStream(const Stream& s, uint32_t first, uint32_t blocks_per_turn) : Stream(s), b0(first), per_turn(blocks_per_turn) { Stream::hi(b0, h0); }
  RT_DEV void hi(uint32_t block, uint32_t h[4]) const {
    if (block == b0) { h[0] = h0[0]; h[1] = h0[1]; h[2] = h0[2]; h[3] = h0[3]; }
    else Stream::hi(block, h);
  }
  RT_DEV uint32_t first_block() const { return b0; }
  RT_DEV bool out_of_turn(uint32_t b) const {
    if (b < b0 + per_turn) return false;
    resume = b;
    return true;
  }
};

// rand 0.8.5 value conversions: Standard f64 keeps 53 bits; gen_range builds a 52-bit mantissa in [1,2) - 1 and scales it
// (a result that rounds up to `hi`, probability about 2^-52, takes the largest value below hi).
RT_DEV float next_toward(float x, float lo) { return x == lo ? x : __uint_as_float(__float_as_uint(x) + ((x > lo) == (x > 0.0f) ? -1 : 1)); }
RT_DEV double next_toward(double x, double lo) {
  return x == lo ? x : __longlong_as_double(__double_as_longlong(x) + ((x > lo) == (x > 0.0) ? -1ll : 1ll));
}
RT_DEV float u01_of(uint32_t h, uint32_t, float) { return (float)(h >> 8) * 0x1.0p-24f; }
// (h << 32 | l) >> 11 scaled by 2^-53 = h * 2^-32 + (l >> 11) * 2^-53: a 53-bit value, so the one multiply-add is exact
RT_DEV double u01_of(uint32_t h, uint32_t l, double) { return __builtin_fma((double)h, 0x1.0p-32, (double)(l >> 11) * 0x1.0p-53); }
RT_DEV float m01_of(uint32_t h, uint32_t, float) { return (float)(h >> 9) * 0x1.0p-23f; }
RT_DEV double m01_of(uint32_t h, uint32_t l, double) { return __builtin_fma((double)h, 0x1.0p-32, (double)(l >> 12) * 0x1.0p-52); }
template <class R> RT_DEV R range_of(uint32_t h, uint32_t l, R lo, R hi) {  // half-open [lo, hi)
  R res = m01_of(h, l, (R)0) * (hi - lo) + lo;
  return res < hi ? res : next_toward(hi, lo);
}
// [-1, 1): never rounds up to 1, no guard needed
template <class R> RT_DEV R signed_unit_of(uint32_t h, uint32_t l) { return m01_of(h, l, (R)0) * (R)2 + (R)-1; }

// One block of four draws.  The double build fetches the low halves as well.
template <class R> struct Block;
template <> struct Block<float> {
  uint32_t h[4];
  template <class S> RT_DEV Block(const S& s, uint32_t b) { s.hi(b, h); }
  RT_DEV float u01(int j) const { return u01_of(h[j], 0u, 0.0f); }
  RT_DEV float range(int j, float lo, float hi) const { return range_of<float>(h[j], 0u, lo, hi); }
};
template <> struct Block<double> {
  uint32_t h[4], l[4];
  template <class S> RT_DEV Block(const S& s, uint32_t b) { s.hi(b, h); Stream::lo_of(h, l); }
  RT_DEV double u01(int j) const { return u01_of(h[j], l[j], 0.0); }
  RT_DEV double range(int j, double lo, double hi) const { return range_of<double>(h[j], l[j], lo, hi); }
};

// ln(U) for the uniform U in [0,1) at a draw slot (free-path sampling, src/constant_medium.rs:91).  A float cannot represent U
// close to 1, where ln(U) ~ -(1 - U) is tiny, so the float build forms 1 - U exactly from the 53-bit integer and uses log1p there.
RT_DEV uint64_t word_at(const Stream& s, uint32_t slot) {
  uint32_t h[4], l[4];
  s.hi(slot >> 2, h);
  Stream::lo_of(h, l);
  uint32_t j = slot & 3u;
  uint32_t hj = j == 0 ? h[0] : j == 1 ? h[1] : j == 2 ? h[2] : h[3];
  uint32_t lj = j == 0 ? l[0] : j == 1 ? l[1] : j == 2 ? l[2] : l[3];
  return ((uint64_t)hj << 32) | lj;
}
RT_DEV double gen_log01(const Stream& s, uint32_t slot, double) {
  const uint64_t w = word_at(s, slot);
  return rt_log(u01_of((uint32_t)(w >> 32), (uint32_t)w, 0.0));
}
// Only the high word of the draw is generated: U moves by less than 2^-32, ln(U) by less than 2^-32 / U in absolute terms, far
// below the resolution of a float path length, and the second Philox block per medium crossing is saved.
RT_DEV float gen_log01(const Stream& s, uint32_t slot, float) {
  uint32_t h[4];
  s.hi(slot >> 2, h);
  const uint32_t j = slot & 3u;
  const uint32_t hj = j == 0 ? h[0] : j == 1 ? h[1] : j == 2 ? h[2] : h[3];
  if (hj >= 0x80000000u) return log1pf(-((float)(0u - hj) * 0x1.0p-32f));  // 1 - U = (2^32 - hj) / 2^32, exact before the conversion
  return rt_log((float)hj * 0x1.0p-32f);
}

// src/vec3.rs:81-88: first candidate (three draws in [-1,1)) inside the unit ball; candidate c is block c of the stream.
// src/vec3.rs:81-88: first candidate (a point of [-1,1)^3) inside the unit ball.  One Philox block of high halves serves two
// candidates: its words (0,1) and (2,3) are each cut into three 21-bit coordinates.
RT_DEV void low3(uint32_t w0, uint32_t w1, uint32_t l[3]) {  // low words of the three coordinates of candidate (w0, w1)
  const uint32_t m = low_mix(w0, w1);
  l[0] = m * 0x846CA68Bu; l[1] = (m + w1) * 0x9E3779B1u; l[2] = (m ^ w0) * 0x85EBCA77u;
}
RT_DEV void fields21(uint32_t w0, uint32_t w1, uint32_t& xi, uint32_t& yi, uint32_t& zi) {
  // bits 63..43, 42..22 and 21..1 of w0:w1: a shift, a funnel shift + mask, a bit-field extract
  xi = w0 >> 11; yi = __builtin_amdgcn_alignbit(w0, w1, 22) & 0x1FFFFFu; zi = __builtin_amdgcn_ubfe(w1, 1, 21);
}
RT_DEV float coord_of(uint32_t hi21, uint32_t, float) { return (float)hi21 * 0x1.0p-20f - 1.0f; }
// gen_range(-1.0..1.0) on the 52-bit mantissa m = hi << 31 | lo: (m * 2^-52) * 2 + -1.  Every step of that is exact (m has 52
// bits; the result is a multiple of 2^-51 in [-1, 1)), so it equals hi * 2^-20 - 1 + lo * 2^-51 formed by two multiply-adds -
// exact as well, the first a multiple of 2^-20 in [-1, 1) - without the 64-bit integer to double conversion (four instructions).
RT_DEV double coord_of(uint32_t hi21, uint32_t lo31, double) {
  return __builtin_fma((double)lo31, 0x1.0p-51, __builtin_fma((double)hi21, 0x1.0p-20, -1.0));
}
template <class R> RT_DEV V3<R> point_of(uint32_t w0, uint32_t w1, uint32_t l0, uint32_t l1, uint32_t l2) {
  uint32_t xi, yi, zi;
  fields21(w0, w1, xi, yi, zi);
  return V3<R>(coord_of(xi, l0 >> 1, (R)0), coord_of(yi, l1 >> 1, (R)0), coord_of(zi, l2 >> 1, (R)0));
}
template <class S> RT_DEV V3<float> random_in_unit_sphere(const S& s, float) {
  for (uint32_t b = s.first_block();; ++b) {
    if (s.out_of_turn(b)) return V3<float>(0.0f, 0.0f, 1.0f);  // not a draw: the caller sees s.resume and calls again
    uint32_t h[4];
    s.hi(b, h);
    V3<float> p = point_of<float>(h[0], h[1], 0u, 0u, 0u);
#ifdef RT_DIAG_ACCEPT_FIRST  // diagnostic build only: cost of the loop beyond its first candidate (changes the images)
    return p;
#endif
    if (length_squared(p) < 1.0f) return p;
    p = point_of<float>(h[2], h[3], 0u, 0u, 0u);
    if (length_squared(p) < 1.0f) return p;
  }
}
// Double build: the high fields alone decide all but a sliver of candidates (the low fields move a coordinate by less than
// 2^-20, the squared length by less than 5.8e-6), so the low words are formed once, for the candidate that passes that test.
// The screening runs in single precision: the 21-bit fields are exact there, and the rounding of the squared length (below 4e-7)
// is covered by the margin.
RT_DEV bool unit_ball_screen(uint32_t w0, uint32_t w1) { return length_squared(point_of<float>(w0, w1, 0u, 0u, 0u)) < 1.0f + 8e-6f; }
RT_DEV V3<double> unit_ball_point(uint32_t w0, uint32_t w1) {
  uint32_t l[3];
  low3(w0, w1, l);
  return point_of<double>(w0, w1, l[0], l[1], l[2]);
}
// The plain loop, from candidate c on.
template <class S> RT_DEV V3<double> unit_ball_loop(const S& s, uint32_t c) {
  uint32_t h[4] = {0u, 0u, 0u, 0u};
  if (c & 1u) s.hi(c >> 1, h);
  for (;;) {
    uint32_t w0, w1;
    for (;; ++c) {
      if ((c & 1u) == 0u) {
        if (s.out_of_turn(c >> 1)) return V3<double>(0.0, 0.0, 1.0);  // not a draw: the caller sees s.resume and calls again
        s.hi(c >> 1, h);
      }
      w0 = (c & 1u) ? h[2] : h[0];
      w1 = (c & 1u) ? h[3] : h[1];
      if (unit_ball_screen(w0, w1)) break;
    }
    const V3<double> p = unit_ball_point(w0, w1);
    if (length_squared(p) < 1.0) return p;
    ++c;
  }
}
template <class S> RT_DEV V3<double> random_in_unit_sphere(const S& s, double) { return unit_ball_loop(s, 2u * s.first_block()); }
// On a primed stream (the path kernels): the block that is already there gives two candidates, screened side by side in every
// lane; the lanes that found none (a quarter) evaluate the next block and screen its two; then the candidate in double - three
// dozen instructions - is formed ONCE for all lanes.  As a loop over candidates that part runs again in every turn in which some
// lane passes the screening, three or four times per draw.  What is left over (no candidate in two blocks when the turn allows
// more, or a candidate inside the sliver between the screening's bound and the ball) goes through the plain loop.
RT_DEV V3<double> random_in_unit_sphere(const PrimedStream& s, double) {
  const uint32_t b0 = s.b0;
  const bool p0 = unit_ball_screen(s.h0[0], s.h0[1]), p1 = unit_ball_screen(s.h0[2], s.h0[3]);
  uint32_t w0 = p0 ? s.h0[0] : s.h0[2], w1 = p0 ? s.h0[1] : s.h0[3];
  uint32_t c = 2u * b0 + (p0 ? 0u : 1u);
  bool have = p0 | p1;
  bool rest = false;  // go on in the plain loop from candidate c
  if (!have) {
    c = 2u * (b0 + 1u);
    if (s.out_of_turn(b0 + 1u)) return V3<double>(0.0, 0.0, 1.0);  // not a draw: the caller sees s.resume and calls again
    uint32_t h[4];
    s.Stream::hi(b0 + 1u, h);
    const bool p2 = unit_ball_screen(h[0], h[1]), p3 = unit_ball_screen(h[2], h[3]);
    w0 = p2 ? h[0] : h[2]; w1 = p2 ? h[1] : h[3];
    c += p2 ? 0u : 1u;
    have = p2 | p3;
    if (!have) { c = 2u * (b0 + 2u); rest = true; }
  }
  V3<double> p(0.0, 0.0, 1.0);
  if (have) {
    p = unit_ball_point(w0, w1);
    if (!(length_squared(p) < 1.0)) { c += 1u; rest = true; }
  }
  if (rest) p = unit_ball_loop(s, c);
  return p;
}
template <class R, class S> RT_DEV V3<R> random_in_unit_sphere(const S& s) { return random_in_unit_sphere(s, (R)0); }
// src/vec3.rs:90-92.  Single precision scales by the hardware reciprocal square root (1 ulp) instead of sqrt then divide.
// RELAXED (double precision, small-scene presets): the refined reciprocal square root (2 ulp) instead of sqrt then divide.
template <bool RELAXED, class S> RT_DEV V3<double> random_unit_vector(const S& s, double) {
  const V3<double> p = random_in_unit_sphere<double>(s);
  if constexpr (RELAXED) return rsq_one_step(length_squared(p)) * p;
  else return unit_vector(p);
}
template <bool RELAXED, class S> RT_DEV V3<float> random_unit_vector(const S& s, float) {
  V3<float> p = random_in_unit_sphere<float>(s);
  return __builtin_amdgcn_rsqf(length_squared(p)) * p;
}
template <class R, bool RELAXED, class S> RT_DEV V3<R> random_unit_vector(const S& s) { return random_unit_vector<RELAXED>(s, (R)0); }
// src/vec3.rs:103-111 on the camera stream: candidate c is block 1 + c.
template <class R, class S> RT_DEV V3<R> random_in_unit_disk(const S& s) {
  for (uint32_t c = 0;; ++c) {
    Block<R> b(s, 1u + c);
    V3<R> p(b.range(0, (R)-1, (R)1), b.range(1, (R)-1, (R)1), (R)0);
    if (length_squared(p) < (R)1) return p;
  }
}

}  // namespace rtd
