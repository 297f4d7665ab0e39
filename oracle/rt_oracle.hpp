// CPU ORACLE — TEST INFRASTRUCTURE ONLY.
//
// A C++17 restatement, function by function, of the path-tracing hot path of
// soikkea/rust-raytracing (Rust, all arithmetic in f64).  It exists to CHECK the
// HIP implementation: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may build, link or call it.  Nothing under
// rust-raytracing_amd/ includes or links this file.
//
// Structure deliberately mirrors the reference (pointer-based `dyn Hittable`
// tree, recursive BVH and ray_color) so it can be audited against the Rust by
// eye; every function cites the reference file:line it follows.
//
// Parity status: the reference's own test suite holds ONE vector for this path
// (aabb.rs:70-83) which this oracle reproduces (tests/test_oracle_golden.py).
// Everything the reference draws from `rand::thread_rng()` (OS-seeded ChaCha12)
// is non-reproducible by design, so the oracle draws from a counter-based
// Philox4x32-7 stream instead (same distributions; every draw has a fixed
// address in the stream of its (pixel, sample, bounce), see Stream).  All
// random-number-dependent outputs are therefore "parity unpinned" against the
// Rust binary and pinned only statistically (committed sample renders).
//
// The template parameter R is the arithmetic type: R=double is THE oracle;
// R=float is only used to study single-precision drift.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <optional>
#include <string>
#include <thread>
#include <vector>

namespace orc {

// ---------------------------------------------------------------------------
// Instrumentation (not in the reference): per-thread event counts used to
// derive the algorithmic flops/bytes per ray quoted by the benchmark.
// ---------------------------------------------------------------------------
struct Counters {
  uint64_t rays = 0;            // ray_color calls with depth > 0 (one closest-hit query each)
  uint64_t camera_rays = 0;
  uint64_t aabb = 0;            // Aabb::hit calls
  uint64_t sphere_miss = 0, sphere_hit = 0;
  uint64_t moving_miss = 0, moving_hit = 0;
  uint64_t rect_miss = 0, rect_hit = 0;
  uint64_t translate = 0, rotate = 0, medium = 0;
  uint64_t sc_lambert = 0, sc_metal = 0, sc_dielectric = 0, sc_light = 0, sc_iso = 0;
  uint64_t tx_solid = 0, tx_checker = 0, tx_noise = 0, tx_image = 0;
  uint64_t bg = 0;
  uint64_t rng_words = 0;       // 64-bit random words consumed
  uint64_t depth_exhausted = 0;
  void add(const Counters& o) {
    const uint64_t* s = reinterpret_cast<const uint64_t*>(&o);
    uint64_t* d = reinterpret_cast<uint64_t*>(this);
    for (size_t i = 0; i < sizeof(Counters) / sizeof(uint64_t); ++i) d[i] += s[i];
  }
};
inline thread_local Counters g_cnt;

// ---------------------------------------------------------------------------
// Philox4x32 (Salmon et al., SC'11) — published round function and constants,
// kPhiloxRounds = 7 rounds (the smallest count the paper reports as passing
// BigCrush; Random123 publishes known-answer vectors for 7 and for 10).  Replaces
// rand::thread_rng (reference call sites: render.rs:226, camera.rs:69,
// vec3.rs:20,25,104, constant_medium.rs:50, material.rs:127).
// ---------------------------------------------------------------------------
constexpr int kPhiloxRounds = 7;
inline void philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], int rounds = kPhiloxRounds) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < rounds; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Sub-stream tags: every (pixel, sample, bounce) owns three independent streams.
enum Stage : uint32_t { STAGE_CAMERA = 0, STAGE_MEDIA = 1, STAGE_SCATTER = 2 };

// One random stream = Philox keyed by (pixel, sample); counter =
// (block, bounce | stage<<16, seed_lo, seed_hi).  Draws are ADDRESSED:
// draw slot s is the 64-bit word  H[s&3] << 32 | L[s&3]  where H is Philox block
// s>>2 and L[j] = low_mix(H[j], H[(j+1)&3]) * 0x846CA68B.  (A single-precision
// evaluation only ever needs H.)  Slot assignment per stage:
//   CAMERA   slot 0 = pixel jitter u, 1 = jitter v, 2 = shutter time;
//            lens-disk candidate c uses slots 4(1+c), 4(1+c)+1
//   SCATTER  unit-sphere candidate c (Lambertian, Metal fuzz, Isotropic) takes
//            two words (w0, w1) of H block c>>1 - words 0,1 for even c, 2,3 for
//            odd c - cut into three 21-bit fields, and the three words
//            low3(w0, w1) for the 31 low bits of each coordinate (52-bit
//            mantissas, the resolution of rand's gen_range); Dielectric's one
//            uniform is slot 0
//   MEDIA    the free-path draw of ConstantMedium object k is slot k, plus the
//            visit bits of HitCtx (one draw per object, call and ray, whatever
//            order the tree visits objects in)
// The rejection loops of vec3.rs:81-88,103-111 therefore read "the first
// candidate c = 0, 1, 2, ... inside the unit ball / disk", which is what the
// sequential loop computes.
// Bits of a draw below those Philox supplies: a multiplicative hash of the Philox words (they only place a value inside an
// interval of 2^-32, or 2^-21 for a unit-ball coordinate, of its range).
inline uint32_t low_mix(uint32_t a, uint32_t b) {
  uint32_t m = (a ^ ((b << 16) | (b >> 16))) * 0x7FEB352Du;
  return m ^ (m >> 15);
}
inline void low3(uint32_t w0, uint32_t w1, uint32_t l[3]) {  // low words of the three coordinates of unit-ball candidate (w0, w1)
  const uint32_t m = low_mix(w0, w1);
  l[0] = m * 0x846CA68Bu; l[1] = (m + w1) * 0x9E3779B1u; l[2] = (m ^ w0) * 0x85EBCA77u;
}
struct Stream {
  uint32_t key[2];
  uint32_t c1, c2, c3;
  uint32_t cached = 0xFFFFFFFFu;  // block held in hi/lo
  uint32_t hi[4], lo[4];
  Stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, Stage st) {
    key[0] = pixel; key[1] = sample;
    c1 = bounce | ((uint32_t)st << 16);
    c2 = (uint32_t)seed; c3 = (uint32_t)(seed >> 32);
  }
  void hi_block(uint32_t b, uint32_t out[4]) const {
    uint32_t c[4] = {b, c1, c2, c3};
    philox4x32(c, key, out);
  }
  uint64_t word(uint32_t slot) {
    uint32_t b = slot >> 2, j = slot & 3u;
    if (b != cached) {
      uint32_t ch[4] = {b, c1, c2, c3};
      philox4x32(ch, key, hi);
      for (int k = 0; k < 4; ++k) lo[k] = low_mix(hi[k], hi[(k + 1) & 3]) * 0x846CA68Bu;
      cached = b;
    }
    ++g_cnt.rng_words;
    return ((uint64_t)hi[j] << 32) | lo[j];
  }
};

// Uniform conversions.  rand 0.8.5 semantics (third-party, not under
// /root/reference; restated from its published algorithm, unpinned):
//   Standard f64      = (u64 >> 11) * 2^-53                       in [0,1)
//   gen_range(lo..hi) = ((u64 >> 12) * 2^-52) * (hi-lo) + lo      half-open
// rand redraws when rounding lands on hi (probability about 2^-52); with
// addressed draws that case takes the largest value below hi instead.
// For R=float the top 24 / 23 bits of the same 64-bit word are used so that a
// float run consumes the stream exactly like the double run.
template <class R> struct Uni;
template <> struct Uni<double> {
  static double u01(uint64_t w) { return (double)(w >> 11) * (1.0 / 9007199254740992.0); }
  static double m01(uint64_t w) { return (double)(w >> 12) * (1.0 / 4503599627370496.0); }
};
template <> struct Uni<float> {
  static float u01(uint64_t w) { return (float)(w >> 40) * (1.0f / 16777216.0f); }
  static float m01(uint64_t w) { return (float)(w >> 41) * (1.0f / 8388608.0f); }
};
template <class R> inline R range_of(uint64_t w, R lo, R hi) {
  R res = Uni<R>::m01(w) * (hi - lo) + lo;
  return res < hi ? res : std::nextafter(hi, lo);
}
template <class R> inline R gen01_at(Stream& s, uint32_t slot) { return Uni<R>::u01(s.word(slot)); }
template <class R> inline R gen_range_at(Stream& s, uint32_t slot, R lo, R hi) { return range_of<R>(s.word(slot), lo, hi); }

// ---------------------------------------------------------------------------
// vec3.rs
// ---------------------------------------------------------------------------
template <class R> struct V3 {
  R e[3];
  V3() : e{0, 0, 0} {}
  V3(R a, R b, R c) : e{a, b, c} {}
  R x() const { return e[0]; }
  R y() const { return e[1]; }
  R z() const { return e[2]; }
  R& operator[](int i) { return e[i]; }
  R operator[](int i) const { return e[i]; }
  R length_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }  // vec3.rs:49-51
  R length() const { return std::sqrt(length_squared()); }                       // vec3.rs:45-47
  bool near_zero() const {                                                       // vec3.rs:53-56
    const R s = (R)1e-8;
    return std::fabs(e[0]) < s && std::fabs(e[1]) < s && std::fabs(e[2]) < s;
  }
  R dot(const V3& o) const { return e[0] * o.e[0] + e[1] * o.e[1] + e[2] * o.e[2]; }  // vec3.rs:58-60
  V3 cross(const V3& o) const {                                                  // vec3.rs:62-68
    return V3(e[1] * o.e[2] - e[2] * o.e[1], e[2] * o.e[0] - e[0] * o.e[2], e[0] * o.e[1] - e[1] * o.e[0]);
  }
};
template <class R> inline V3<R> operator+(const V3<R>& a, const V3<R>& b) { return {a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]}; }
template <class R> inline V3<R> operator-(const V3<R>& a, const V3<R>& b) { return {a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]}; }
template <class R> inline V3<R> operator*(const V3<R>& a, const V3<R>& b) { return {a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]}; }
template <class R> inline V3<R> operator*(R f, const V3<R>& v) { return {v.e[0] * f, v.e[1] * f, v.e[2] * f}; }  // vec3.rs:200-206
template <class R> inline V3<R> operator*(const V3<R>& v, R f) { return f * v; }
template <class R> inline V3<R> operator/(const V3<R>& v, R f) { return ((R)1 / f) * v; }                    // vec3.rs:224-230
template <class R> inline V3<R> operator-(const V3<R>& v) { return {-v.e[0], -v.e[1], -v.e[2]}; }
template <class R> inline V3<R> unit_vector(const V3<R>& v) { return v / v.length(); }                         // vec3.rs:77-79

// Coordinate in [-1, 1) from a 21-bit high field and a 31-bit low field: gen_range(-1.0..1.0) on the 52-bit mantissa
// hi << 31 | lo (vec3.rs:24-31).  A float keeps the high field only.
inline double coord_of(uint32_t hi21, uint32_t lo31, double) {
  uint64_t m = ((uint64_t)hi21 << 31) | lo31;
  return ((double)m * (1.0 / 4503599627370496.0)) * 2.0 + -1.0;
}
inline float coord_of(uint32_t hi21, uint32_t, float) { return (float)hi21 * (1.0f / 1048576.0f) - 1.0f; }
template <class R> inline V3<R> random_in_unit_sphere(Stream& s) {  // vec3.rs:81-88: first candidate c = 0, 1, 2, ... inside the unit ball
  uint32_t h[4] = {0, 0, 0, 0}, l[4];
  for (uint32_t c = 0;; ++c) {
    if ((c & 1u) == 0) s.hi_block(c >> 1, h);
    g_cnt.rng_words += 3;
    uint32_t w0 = (c & 1u) ? h[2] : h[0], w1 = (c & 1u) ? h[3] : h[1];
    low3(w0, w1, l);
    uint32_t xi = w0 >> 11, yi = ((w0 & 0x7FFu) << 10) | (w1 >> 22), zi = (w1 >> 1) & 0x1FFFFFu;
    V3<R> p(coord_of(xi, l[0] >> 1, (R)0), coord_of(yi, l[1] >> 1, (R)0), coord_of(zi, l[2] >> 1, (R)0));
    if (p.length_squared() < (R)1) return p;
  }
}
template <class R> inline V3<R> random_unit_vector(Stream& s) { return unit_vector(random_in_unit_sphere<R>(s)); }  // vec3.rs:90-92
template <class R> inline V3<R> random_in_unit_disk(Stream& s) {  // vec3.rs:103-111; candidate c = block 1 + c of the camera stream
  for (uint32_t c = 0;; ++c) {
    R a = gen_range_at<R>(s, 4 * (1 + c) + 0, (R)-1, (R)1);
    R b = gen_range_at<R>(s, 4 * (1 + c) + 1, (R)-1, (R)1);
    V3<R> p(a, b, 0);
    if (p.length_squared() < (R)1) return p;
  }
}
template <class R> inline V3<R> reflect(const V3<R>& v, const V3<R>& n) { return v - ((R)2 * v.dot(n)) * n; }  // vec3.rs:113-115
template <class R> inline V3<R> refract(const V3<R>& uv, const V3<R>& n, R etai_over_etat) {                 // vec3.rs:117-122
  R cos_theta = std::fmin((-uv).dot(n), (R)1);
  V3<R> r_out_perp = etai_over_etat * (uv + cos_theta * n);
  V3<R> r_out_parallel = (-std::sqrt(std::fabs((R)1 - r_out_perp.length_squared()))) * n;
  return r_out_perp + r_out_parallel;
}

// ---------------------------------------------------------------------------
// ray.rs
// ---------------------------------------------------------------------------
template <class R> struct Ray {
  V3<R> origin, direction;
  R time = 0;
  V3<R> at(R t) const { return origin + (t * direction); }  // ray.rs:31-33
};

// Per-query context: the MEDIA random stream of the current (pixel,sample,bounce).
// `visit`: which call of a repeated object this is.  A one-object BVH node holds the same child twice (bvh.rs:38) and hits it
// twice (bvh.rs:114-116); the reference's ConstantMedium draws a fresh random number on every call (constant_medium.rs:50,91), so
// with addressed draws the second call needs a slot of its own: bit 0 = second call in a node of the world's tree, bit 1 =
// second call in a node of a nested tree.
// `skip_*`: test tooling, not part of the reference.  The runner-up query of the parity tests repeats a closest-hit query with
// the winning surface left out: a leaf whose object belongs to class `skip_class` (classes group exact copies of an object) and
// whose nested index and box side equal skip_sub / skip_side reports a miss.
struct HitCtx {
  Stream* media;
  uint32_t visit = 0;
  const int32_t* class_of_obj = nullptr;
  int skip_class = -1, skip_sub = -1, skip_side = -1;
  bool skipped(int obj, int sub, int side) const {
    return class_of_obj && obj >= 0 && class_of_obj[obj] == skip_class && sub == skip_sub && side == skip_side;
  }
};

// ---------------------------------------------------------------------------
// aabb.rs
// ---------------------------------------------------------------------------
template <class R> struct Aabb {
  V3<R> minimum, maximum;
  bool hit(const Ray<R>& ray, R t_min, R t_max) const {  // aabb.rs:30-46
    ++g_cnt.aabb;
    for (int a = 0; a < 3; ++a) {
      R inv_d = (R)1 / ray.direction[a];
      R t0 = (minimum[a] - ray.origin[a]) * inv_d;
      R t1 = (maximum[a] - ray.origin[a]) * inv_d;
      if (inv_d < (R)0) std::swap(t0, t1);
      t_min = t0 > t_min ? t0 : t_min;
      t_max = t1 < t_max ? t1 : t_max;
      if (t_max <= t_min) return false;
    }
    return true;
  }
  Aabb surrounding_box(const Aabb& o) const {  // aabb.rs:48-61
    V3<R> small(std::fmin(minimum.x(), o.minimum.x()), std::fmin(minimum.y(), o.minimum.y()), std::fmin(minimum.z(), o.minimum.z()));
    V3<R> big(std::fmax(maximum.x(), o.maximum.x()), std::fmax(maximum.y(), o.maximum.y()), std::fmax(maximum.z(), o.maximum.z()));
    return {small, big};
  }
};

// ---------------------------------------------------------------------------
// texture.rs / perlin.rs
// ---------------------------------------------------------------------------
template <class R> struct Texture {
  virtual ~Texture() = default;
  virtual V3<R> value(R u, R v, const V3<R>& p) const = 0;
};
template <class R> struct SolidColor : Texture<R> {
  V3<R> c;
  explicit SolidColor(V3<R> c_) : c(c_) {}
  V3<R> value(R, R, const V3<R>&) const override { ++g_cnt.tx_solid; return c; }  // texture.rs:29-33
};
template <class R> struct CheckerTexture : Texture<R> {
  std::shared_ptr<Texture<R>> odd, even;
  V3<R> value(R u, R v, const V3<R>& p) const override {  // texture.rs:54-63
    ++g_cnt.tx_checker;
    R sines = std::sin((R)10 * p.x()) * std::sin((R)10 * p.y()) * std::sin((R)10 * p.z());
    return sines < (R)0 ? odd->value(u, v, p) : even->value(u, v, p);
  }
};
template <class R> struct Perlin {
  V3<R> ran_vec[256];
  int perm_x[256], perm_y[256], perm_z[256];
  R noise(const V3<R>& pt) const {  // perlin.rs:32-55
    R u = pt.x() - std::floor(pt.x());
    R v = pt.y() - std::floor(pt.y());
    R w = pt.z() - std::floor(pt.z());
    int i = (int)std::floor(pt.x());
    int j = (int)std::floor(pt.y());
    int k = (int)std::floor(pt.z());
    V3<R> c[2][2][2];
    for (int di = 0; di < 2; ++di)
      for (int dj = 0; dj < 2; ++dj)
        for (int dk = 0; dk < 2; ++dk)
          c[di][dj][dk] = ran_vec[perm_x[(i + di) & 255] ^ perm_y[(j + dj) & 255] ^ perm_z[(k + dk) & 255]];
    return interp(c, u, v, w);
  }
  static R interp(const V3<R> c[2][2][2], R u, R v, R w) {  // perlin.rs:116-139
    R uu = u * u * ((R)3 - (R)2 * u);
    R vv = v * v * ((R)3 - (R)2 * v);
    R ww = w * w * ((R)3 - (R)2 * w);
    R accum = 0;
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        for (int k = 0; k < 2; ++k) {
          R fi = (R)i, fj = (R)j, fk = (R)k;
          V3<R> weight_v(u - fi, v - fj, w - fk);
          accum += (fi * uu + ((R)1 - fi) * ((R)1 - uu)) * (fj * vv + ((R)1 - fj) * ((R)1 - vv)) *
                   (fk * ww + ((R)1 - fk) * ((R)1 - ww)) * c[i][j][k].dot(weight_v);
        }
    return accum;
  }
  R turbulence(const V3<R>& pt, int depth) const {  // perlin.rs:60-72
    R accum = 0;
    V3<R> temp_p = pt;
    R weight = 1;
    for (int i = 0; i < depth; ++i) {
      accum += weight * noise(temp_p);
      weight *= (R)0.5;
      temp_p = (R)2 * temp_p;
    }
    return std::fabs(accum);
  }
};
template <class R> struct NoiseTexture : Texture<R> {
  Perlin<R> noise;
  R scale;
  V3<R> value(R, R, const V3<R>& p) const override {  // texture.rs:79-85
    ++g_cnt.tx_noise;
    return (V3<R>(1, 1, 1) * (R)0.5) * ((R)1 + std::sin(scale * p.z() + (R)10 * noise.turbulence(p, 7)));
  }
};
template <class R> struct ImageTexture : Texture<R> {
  std::vector<uint8_t> data;
  uint32_t width = 0, height = 0, bytes_per_scanline = 0;
  V3<R> value(R u, R v, const V3<R>&) const override {  // texture.rs:128-159
    ++g_cnt.tx_image;
    if (data.empty()) return V3<R>(0, 1, 1);
    u = std::fmin(std::fmax(u, (R)0), (R)1);
    v = (R)1 - std::fmin(std::fmax(v, (R)0), (R)1);
    uint32_t i = sat_u32(u * (R)width);
    uint32_t j = sat_u32(v * (R)height);
    if (i >= width) i = width - 1;
    if (j >= height) j = height - 1;
    R color_scale = (R)1 / (R)255;
    size_t px = (size_t)j * bytes_per_scanline + (size_t)i * 3;
    return V3<R>(color_scale * (R)data[px], color_scale * (R)data[px + 1], color_scale * (R)data[px + 2]);
  }
  static uint32_t sat_u32(R x) {  // Rust `as u32`: saturating, NaN -> 0
    if (!(x > (R)0)) return 0;
    if (x >= (R)4294967295.0) return 4294967295u;
    return (uint32_t)x;
  }
};

// ---------------------------------------------------------------------------
// hittable.rs: HitRecord
// ---------------------------------------------------------------------------
template <class R> struct Material;
template <class R> struct HitRecord {  // hittable.rs:10-18
  V3<R> p, normal;
  const Material<R>* material = nullptr;
  R t = 0;
  bool front_face = false;
  R u = 0, v = 0;
  // Not in the reference: identity of the shape that produced the record
  // (top-level world index, index inside a nested list or -1, box side or -1).
  int obj = -1, sub = -1, side = -1;
  void set_face_normal(const Ray<R>& ray, const V3<R>& outward_normal) {  // hittable.rs:79-86
    front_face = ray.direction.dot(outward_normal) < (R)0;
    normal = front_face ? outward_normal : -outward_normal;
  }
};

// ---------------------------------------------------------------------------
// material.rs
// ---------------------------------------------------------------------------
template <class R> struct ScatterResult {
  V3<R> attenuation;
  Ray<R> scattered;
};
template <class R> struct Material {
  int id = -1;
  virtual ~Material() = default;
  virtual std::optional<ScatterResult<R>> scatter(const Ray<R>& ray_in, const HitRecord<R>& rec, Stream& rng) const = 0;
  virtual V3<R> emitted(R, R, const V3<R>&) const { return V3<R>(); }  // material.rs:18-20
};
template <class R> struct Lambertian : Material<R> {
  std::shared_ptr<Texture<R>> albedo;
  std::optional<ScatterResult<R>> scatter(const Ray<R>& ray_in, const HitRecord<R>& rec, Stream& rng) const override {  // material.rs:43-57
    ++g_cnt.sc_lambert;
    V3<R> scatter_direction = rec.normal + random_unit_vector<R>(rng);
    if (scatter_direction.near_zero()) scatter_direction = rec.normal;
    Ray<R> scattered{rec.p, scatter_direction, ray_in.time};
    V3<R> attenuation = albedo->value(rec.u, rec.v, rec.p);
    return ScatterResult<R>{attenuation, scattered};
  }
};
template <class R> struct Metal : Material<R> {
  V3<R> albedo;
  R fuzz;
  Metal(V3<R> a, R f) : albedo(a), fuzz(f < (R)1 ? f : (R)1) {}  // material.rs:66-71
  std::optional<ScatterResult<R>> scatter(const Ray<R>& ray_in, const HitRecord<R>& rec, Stream& rng) const override {  // material.rs:75-90
    ++g_cnt.sc_metal;
    V3<R> reflected = reflect(unit_vector(ray_in.direction), rec.normal);
    Ray<R> scattered{rec.p, reflected + fuzz * random_in_unit_sphere<R>(rng), ray_in.time};
    if (scattered.direction.dot(rec.normal) > (R)0) return ScatterResult<R>{albedo, scattered};
    return std::nullopt;
  }
};
template <class R> struct Dielectric : Material<R> {
  R ir;
  explicit Dielectric(R i) : ir(i) {}
  static R reflectance(R cosine, R ref_idx) {  // material.rs:104-109 (powi(5) = x * (x^2)^2)
    R r0 = ((R)1 - ref_idx) / ((R)1 + ref_idx);
    r0 = r0 * r0;
    R x = (R)1 - cosine;
    R x2 = x * x;
    R x4 = x2 * x2;
    return r0 + ((R)1 - r0) * (x * x4);
  }
  std::optional<ScatterResult<R>> scatter(const Ray<R>& ray_in, const HitRecord<R>& rec, Stream& rng) const override {  // material.rs:113-139
    ++g_cnt.sc_dielectric;
    V3<R> attenuation(1, 1, 1);
    R refraction_ratio = rec.front_face ? (R)1 / ir : ir;
    V3<R> unit_direction = unit_vector(ray_in.direction);
    R cos_theta = std::fmin((-unit_direction).dot(rec.normal), (R)1);
    R sin_theta = std::sqrt((R)1 - cos_theta * cos_theta);
    bool cannot_refract = refraction_ratio * sin_theta > (R)1;
    V3<R> direction;
    // `||` short-circuits: the random number is drawn only when refraction is possible.
    if (cannot_refract || reflectance(cos_theta, refraction_ratio) > gen01_at<R>(rng, 0))
      direction = reflect(unit_direction, rec.normal);
    else
      direction = refract(unit_direction, rec.normal, refraction_ratio);
    return ScatterResult<R>{attenuation, Ray<R>{rec.p, direction, ray_in.time}};
  }
};
template <class R> struct DiffuseLight : Material<R> {
  std::shared_ptr<Texture<R>> emit;
  V3<R> emitted(R u, R v, const V3<R>& p) const override { return emit->value(u, v, p); }  // material.rs:160-162
  std::optional<ScatterResult<R>> scatter(const Ray<R>&, const HitRecord<R>&, Stream&) const override {  // material.rs:164-166
    ++g_cnt.sc_light;
    return std::nullopt;
  }
};
template <class R> struct Isotropic : Material<R> {
  std::shared_ptr<Texture<R>> albedo;
  std::optional<ScatterResult<R>> scatter(const Ray<R>& ray_in, const HitRecord<R>& rec, Stream& rng) const override {  // material.rs:187-194
    ++g_cnt.sc_iso;
    Ray<R> scattered{rec.p, random_in_unit_sphere<R>(rng), ray_in.time};
    V3<R> attenuation = albedo->value(rec.u, rec.v, rec.p);
    return ScatterResult<R>{attenuation, scattered};
  }
};

// ---------------------------------------------------------------------------
// hittable.rs: trait + Translate + RotateY
// ---------------------------------------------------------------------------
template <class R> struct Hittable {
  virtual ~Hittable() = default;
  virtual bool hit(const Ray<R>& ray, R t_min, R t_max, HitRecord<R>& rec, HitCtx& ctx) const = 0;
  virtual std::optional<Aabb<R>> bounding_box(R time0, R time1) const = 0;
  // Not in the reference: label the shapes below this node (see HitRecord::obj/sub).
  virtual void label(int obj, int sub) = 0;
};
template <class R> using HPtr = std::shared_ptr<Hittable<R>>;
constexpr int KEEP_SUB = -2;

template <class R> struct Translate : Hittable<R> {
  HPtr<R> child;
  V3<R> offset;
  Translate(HPtr<R> c, V3<R> o) : child(std::move(c)), offset(o) {}
  std::optional<Aabb<R>> bounding_box(R t0, R t1) const override {  // hittable.rs:127-138
    auto b = child->bounding_box(t0, t1);
    if (!b) return std::nullopt;
    return Aabb<R>{b->minimum + offset, b->maximum + offset};
  }
  bool hit(const Ray<R>& ray, R t_min, R t_max, HitRecord<R>& rec, HitCtx& ctx) const override {  // hittable.rs:140-151
    ++g_cnt.translate;
    Ray<R> moved{ray.origin - offset, ray.direction, ray.time};
    if (!child->hit(moved, t_min, t_max, rec, ctx)) return false;
    rec.p = rec.p + offset;
    V3<R> outward = rec.normal;
    rec.set_face_normal(moved, outward);
    return true;
  }
  void label(int obj, int sub) override { child->label(obj, sub); }
};

template <class R> struct RotateY : Hittable<R> {
  HPtr<R> child;
  R sin_theta, cos_theta;
  std::optional<Aabb<R>> bbox;
  RotateY(HPtr<R> c, R angle_degrees) : child(std::move(c)) {  // hittable.rs:162-205
    R radians = angle_degrees * (R)(3.14159265358979323846264338327950288 / 180.0);
    sin_theta = std::sin(radians);
    cos_theta = std::cos(radians);
    bbox = child->bounding_box(0, 1);
    Aabb<R> b = bbox ? *bbox : Aabb<R>{};
    const R inf = std::numeric_limits<R>::infinity();
    V3<R> mn(inf, inf, inf), mx(-inf, -inf, -inf);
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        for (int k = 0; k < 2; ++k) {
          R x = (R)i * b.maximum.x() + (R)(1 - i) * b.minimum.x();
          R y = (R)j * b.maximum.y() + (R)(1 - j) * b.minimum.y();
          R z = (R)k * b.maximum.z() + (R)(1 - k) * b.minimum.z();
          R new_x = cos_theta * x + sin_theta * z;
          R new_z = -sin_theta * x + cos_theta * z;
          V3<R> tester(new_x, y, new_z);
          for (int c2 = 0; c2 < 3; ++c2) {
            mn[c2] = std::fmin(mn[c2], tester[c2]);
            mx[c2] = std::fmax(mx[c2], tester[c2]);
          }
        }
    if (bbox) bbox = Aabb<R>{mn, mx};
  }
  std::optional<Aabb<R>> bounding_box(R, R) const override { return bbox; }  // hittable.rs:209-211
  bool hit(const Ray<R>& ray, R t_min, R t_max, HitRecord<R>& rec, HitCtx& ctx) const override {  // hittable.rs:213-242
    ++g_cnt.rotate;
    V3<R> origin = ray.origin, direction = ray.direction;
    origin[0] = cos_theta * ray.origin[0] - sin_theta * ray.origin[2];
    origin[2] = sin_theta * ray.origin[0] + cos_theta * ray.origin[2];
    direction[0] = cos_theta * ray.direction[0] - sin_theta * ray.direction[2];
    direction[2] = sin_theta * ray.direction[0] + cos_theta * ray.direction[2];
    Ray<R> rotated{origin, direction, (R)0};  // time forced to 0.0 (hittable.rs:223)
    if (!child->hit(rotated, t_min, t_max, rec, ctx)) return false;
    V3<R> p = rec.p, normal = rec.normal;
    p[0] = cos_theta * rec.p[0] + sin_theta * rec.p[2];
    p[2] = -sin_theta * rec.p[0] + cos_theta * rec.p[2];
    normal[0] = cos_theta * rec.normal[0] + sin_theta * rec.normal[2];
    normal[2] = -sin_theta * rec.normal[0] + cos_theta * rec.normal[2];
    rec.p = p;
    rec.set_face_normal(rotated, normal);
    return true;
  }
  void label(int obj, int sub) override { child->label(obj, sub); }
};

// ---------------------------------------------------------------------------
// sphere.rs / moving_sphere.rs
// ---------------------------------------------------------------------------
template <class R> inline void get_sphere_uv(const V3<R>& p, R& u, R& v) {  // sphere.rs:24-34
  const R PI = (R)3.14159265358979323846264338327950288;
  R theta = std::acos(-p.y());
  R phi = std::atan2(-p.z(), p.x()) + PI;
  u = phi / ((R)2 * PI);
  v = theta / PI;
}
template <class R> struct Sphere : Hittable<R> {
  V3<R> center;
  R radius;
  std::shared_ptr<Material<R>> material;
  int obj = -1, sub = -1;
  Sphere(V3<R> c, R r, std::shared_ptr<Material<R>> m) : center(c), radius(r), material(std::move(m)) {}
  bool hit(const Ray<R>& ray, R t_min, R t_max, HitRecord<R>& rec, HitCtx& ctx) const override {  // sphere.rs:38-73
    if (ctx.skipped(obj, sub, -1)) return false;
    V3<R> oc = ray.origin - center;
    R a = ray.direction.length_squared();
    R half_b = oc.dot(ray.direction);
    R c = oc.length_squared() - radius * radius;
    R discriminant = half_b * half_b - a * c;
    if (discriminant < (R)0) { ++g_cnt.sphere_miss; return false; }
    R sqrt_d = std::sqrt(discriminant);
    R root = (-half_b - sqrt_d) / a;
    if (root < t_min || t_max < root) {
      root = (-half_b + sqrt_d) / a;
      if (root < t_min || t_max < root) { ++g_cnt.sphere_miss; return false; }
    }
    ++g_cnt.sphere_hit;
    V3<R> p = ray.at(root);
    V3<R> outward_normal = (p - center) / radius;
    rec.t = root;
    rec.p = p;
    rec.set_face_normal(ray, outward_normal);
    rec.material = material.get();
    get_sphere_uv(outward_normal, rec.u, rec.v);
    rec.obj = obj; rec.sub = sub; rec.side = -1;
    return true;
  }
  std::optional<Aabb<R>> bounding_box(R, R) const override {  // sphere.rs:75-81
    return Aabb<R>{center - V3<R>(radius, radius, radius), center + V3<R>(radius, radius, radius)};
  }
  void label(int o, int s) override { obj = o; if (s != KEEP_SUB) sub = s; }
};
template <class R> struct MovingSphere : Hittable<R> {
  V3<R> center0, center1;
  R radius;
  std::shared_ptr<Material<R>> material;
  R time0, time1;
  int obj = -1, sub = -1;
  MovingSphere(V3<R> c0, V3<R> c1, R t0, R t1, R r, std::shared_ptr<Material<R>> m)
      : center0(c0), center1(c1), radius(r), material(std::move(m)), time0(t0), time1(t1) {}
  V3<R> center(R time) const { return center0 + ((time - time0) / (time1 - time0)) * (center1 - center0); }  // moving_sphere.rs:37-40
  bool hit(const Ray<R>& ray, R t_min, R t_max, HitRecord<R>& rec, HitCtx& ctx) const override {  // moving_sphere.rs:44-72
    if (ctx.skipped(obj, sub, -1)) return false;
    V3<R> oc = ray.origin - center(ray.time);
    R a = ray.direction.length_squared();
    R half_b = oc.dot(ray.direction);
    R c = oc.length_squared() - radius * radius;
    R discriminant = half_b * half_b - a * c;
    if (discriminant < (R)0) { ++g_cnt.moving_miss; return false; }
    R sqrt_d = std::sqrt(discriminant);
    R root = (-half_b - sqrt_d) / a;
    if (root < t_min || t_max < root) {
      root = (-half_b + sqrt_d) / a;
      if (root < t_min || t_max < root) { ++g_cnt.moving_miss; return false; }
    }
    ++g_cnt.moving_hit;
    V3<R> p = ray.at(root);
    V3<R> outward_normal = (p - center(ray.time)) / radius;
    rec.t = root;
    rec.p = p;
    rec.set_face_normal(ray, outward_normal);
    rec.material = material.get();  // u, v deliberately untouched (moving_sphere.rs:66-69)
    rec.obj = obj; rec.sub = sub; rec.side = -1;
    return true;
  }
  std::optional<Aabb<R>> bounding_box(R t0, R t1) const override {  // moving_sphere.rs:74-85
    V3<R> rv(radius, radius, radius);
    Aabb<R> b0{center(t0) - rv, center(t0) + rv};
    Aabb<R> b1{center(t1) - rv, center(t1) + rv};
    return b0.surrounding_box(b1);
  }
  void label(int o, int s) override { obj = o; if (s != KEEP_SUB) sub = s; }
};

// ---------------------------------------------------------------------------
// aarect.rs — AXIS = the constant axis (2: XYRect, 1: XZRect, 0: YZRect)
// ---------------------------------------------------------------------------
template <class R, int AXIS> struct Rect : Hittable<R> {
  // XYRect: (a,b) = (x,y); XZRect: (a,b) = (x,z); YZRect: (a,b) = (y,z)
  static constexpr int A = (AXIS == 0) ? 1 : 0;
  static constexpr int B = (AXIS == 2) ? 1 : 2;
  std::shared_ptr<Material<R>> material;
  R a0, a1, b0, b1, k;
  int obj = -1, sub = -1, side = -1;
  Rect(R a0_, R a1_, R b0_, R b1_, R k_, std::shared_ptr<Material<R>> m) : material(std::move(m)), a0(a0_), a1(a1_), b0(b0_), b1(b1_), k(k_) {}
  std::optional<Aabb<R>> bounding_box(R, R) const override {  // aarect.rs:76-86,111-121,146-156
    const R padding = (R)0.0001;
    V3<R> mn, mx;
    mn[A] = a0; mx[A] = a1; mn[B] = b0; mx[B] = b1;
    mn[AXIS] = k - padding; mx[AXIS] = k + padding;
    return Aabb<R>{mn, mx};
  }
  bool hit(const Ray<R>& ray, R t_min, R t_max, HitRecord<R>& rec, HitCtx& ctx) const override {  // aarect.rs:88-107,123-142,158-177
    if (ctx.skipped(obj, sub, side)) return false;
    R t = (k - ray.origin[AXIS]) / ray.direction[AXIS];
    if (t < t_min || t > t_max) { ++g_cnt.rect_miss; return false; }
    R a = ray.origin[A] + t * ray.direction[A];
    R b = ray.origin[B] + t * ray.direction[B];
    if (a < a0 || a > a1 || b < b0 || b > b1) { ++g_cnt.rect_miss; return false; }
    ++g_cnt.rect_hit;
    rec.u = (a - a0) / (a1 - a0);
    rec.v = (b - b0) / (b1 - b0);
    rec.t = t;
    V3<R> outward_normal;
    outward_normal[AXIS] = 1;
    rec.set_face_normal(ray, outward_normal);
    rec.material = material.get();
    rec.p = ray.at(t);
    rec.obj = obj; rec.sub = sub; rec.side = side;
    return true;
  }
  void label(int o, int s) override { obj = o; if (s != KEEP_SUB) sub = s; }
};
template <class R> using XYRect = Rect<R, 2>;
template <class R> using XZRect = Rect<R, 1>;
template <class R> using YZRect = Rect<R, 0>;

// ---------------------------------------------------------------------------
// hittable_list.rs
// ---------------------------------------------------------------------------
template <class R> struct HittableList : Hittable<R> {
  std::vector<HPtr<R>> objects;
  void add(HPtr<R> o) { objects.push_back(std::move(o)); }
  bool hit(const Ray<R>& ray, R t_min, R t_max, HitRecord<R>& rec, HitCtx& ctx) const override {  // hittable_list.rs:29-43
    HitRecord<R> temp_rec;
    bool hit_anything = false;
    R closest_so_far = t_max;
    for (const auto& o : objects) {
      if (o->hit(ray, t_min, closest_so_far, temp_rec, ctx)) {
        closest_so_far = temp_rec.t;
        hit_anything = true;
        rec = temp_rec;  // copy_from (hittable.rs:58-69)
      }
    }
    return hit_anything;
  }
  std::optional<Aabb<R>> bounding_box(R t0, R t1) const override {  // hittable_list.rs:45-69
    if (objects.empty()) return std::nullopt;
    std::optional<Aabb<R>> out;
    for (const auto& o : objects) {
      auto b = o->bounding_box(t0, t1);
      if (!b) return std::nullopt;
      out = out ? out->surrounding_box(*b) : *b;
    }
    return out;
  }
  void label(int obj, int sub) override {
    for (auto& o : objects) o->label(obj, sub);
  }
};

// ---------------------------------------------------------------------------
// box_struct.rs
// ---------------------------------------------------------------------------
template <class R> struct BoxShape : Hittable<R> {
  V3<R> mn, mx;
  HittableList<R> sides;
  BoxShape(V3<R> lo, V3<R> hi, std::shared_ptr<Material<R>> m) : mn(lo), mx(hi) {  // box_struct.rs:19-78
    auto add = [&](auto r, int side) { r->side = side; sides.add(r); };
    add(std::make_shared<XYRect<R>>(lo.x(), hi.x(), lo.y(), hi.y(), hi.z(), m), 0);
    add(std::make_shared<XYRect<R>>(lo.x(), hi.x(), lo.y(), hi.y(), lo.z(), m), 1);
    add(std::make_shared<XZRect<R>>(lo.x(), hi.x(), lo.z(), hi.z(), hi.y(), m), 2);
    add(std::make_shared<XZRect<R>>(lo.x(), hi.x(), lo.z(), hi.z(), lo.y(), m), 3);
    add(std::make_shared<YZRect<R>>(lo.y(), hi.y(), lo.z(), hi.z(), hi.x(), m), 4);
    add(std::make_shared<YZRect<R>>(lo.y(), hi.y(), lo.z(), hi.z(), lo.x(), m), 5);
  }
  std::optional<Aabb<R>> bounding_box(R, R) const override { return Aabb<R>{mn, mx}; }  // box_struct.rs:82-84
  bool hit(const Ray<R>& ray, R t_min, R t_max, HitRecord<R>& rec, HitCtx& ctx) const override {  // box_struct.rs:86-94
    return sides.hit(ray, t_min, t_max, rec, ctx);
  }
  void label(int obj, int sub) override { sides.label(obj, sub); }
};

// ---------------------------------------------------------------------------
// constant_medium.rs
// ---------------------------------------------------------------------------
template <class R> struct ConstantMedium : Hittable<R> {
  HPtr<R> boundary;
  std::shared_ptr<Material<R>> phase_function;
  R negative_inverse_density;
  int obj = -1, sub = -1;
  ConstantMedium(HPtr<R> b, R density, std::shared_ptr<Material<R>> phase)
      : boundary(std::move(b)), phase_function(std::move(phase)), negative_inverse_density((R)-1 / density) {}  // constant_medium.rs:19-33
  std::optional<Aabb<R>> bounding_box(R t0, R t1) const override { return boundary->bounding_box(t0, t1); }  // :37-39
  bool hit(const Ray<R>& ray, R t_min, R t_max, HitRecord<R>& rec, HitCtx& ctx) const override {  // constant_medium.rs:41-111
    if (ctx.skipped(obj, sub, -1)) return false;
    ++g_cnt.medium;
    const R inf = std::numeric_limits<R>::infinity();
    HitRecord<R> rec1, rec2;
    if (!boundary->hit(ray, -inf, inf, rec1, ctx)) return false;
    if (!boundary->hit(ray, rec1.t + (R)0.0001, inf, rec2, ctx)) return false;
    if (rec1.t < t_min) rec1.t = t_min;
    if (rec2.t > t_max) rec2.t = t_max;
    if (rec1.t >= rec2.t) return false;
    if (rec1.t < (R)0) rec1.t = 0;
    R ray_length = ray.direction.length();
    R distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
    // draw slot = this object's index, plus the visit bits (an independent draw per call, as thread_rng() gives the reference)
    R hit_distance = negative_inverse_density * std::log(gen01_at<R>(*ctx.media, (uint32_t)obj | (ctx.visit << 28)));
    if (hit_distance > distance_inside_boundary) return false;
    rec.t = rec1.t + hit_distance / ray_length;
    rec.p = ray.at(rec.t);
    rec.normal = V3<R>(1, 0, 0);
    rec.front_face = true;
    rec.material = phase_function.get();  // u, v untouched
    rec.obj = obj; rec.sub = sub; rec.side = -1;
    return true;
  }
  void label(int o, int s) override { obj = o; if (s != KEEP_SUB) sub = s; }  // the boundary's own labels are irrelevant (rec1/rec2 are discarded)
};

// ---------------------------------------------------------------------------
// Seeded host-side generator for scene construction and BVH axes.
// PCG-XSH-RR 64/32 as published (O'Neill 2014), seeded the way rand_core 0.6.4
// SeedableRng::seed_from_u64 + rand_pcg 0.3.1 Lcg64Xsh32::from_seed do
// (reference call sites scenes.rs:158,439).  Third-party crates are absent
// from /root/reference: restated from the published algorithm, unpinned.
// ---------------------------------------------------------------------------
struct Pcg32 {
  uint64_t state = 0, increment = 1;
  static constexpr uint64_t MUL = 6364136223846793005ull;
  static Pcg32 seed_from_u64(uint64_t s) {
    uint32_t w[4];
    for (int i = 0; i < 4; ++i) {
      s = s * MUL + 11634580027462260723ull;
      uint32_t xorshifted = (uint32_t)(((s >> 18) ^ s) >> 27);
      uint32_t rot = (uint32_t)(s >> 59);
      w[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    Pcg32 g;
    uint64_t st = ((uint64_t)w[1] << 32) | w[0];
    uint64_t inc = ((uint64_t)w[3] << 32) | w[2];
    g.increment = inc | 1;
    g.state = st + g.increment;
    g.step();
    return g;
  }
  void step() { state = state * MUL + increment; }
  uint32_t next_u32() {
    uint64_t s = state;
    step();
    uint32_t rot = (uint32_t)(s >> 59);
    uint32_t xsh = (uint32_t)(((s >> 18) ^ s) >> 27);
    return (xsh >> rot) | (xsh << ((32 - rot) & 31));
  }
  uint64_t next_u64() {
    uint64_t x = next_u32();
    uint64_t y = next_u32();
    return (y << 32) | x;
  }
  double gen_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
  double gen_range_f64(double lo, double hi) {
    double scale = hi - lo;
    for (;;) {
      double res = ((double)(next_u64() >> 12) * (1.0 / 4503599627370496.0)) * scale + lo;
      if (res < hi) return res;
      scale = std::nextafter(scale, 0.0);
    }
  }
  // rand 0.8.5 UniformInt::sample_single_inclusive for 32-bit types (widening multiply + zone rejection).
  uint32_t gen_range_u32_inclusive(uint32_t lo, uint32_t hi) {
    uint32_t range = hi - lo + 1;
    if (range == 0) return next_u32();
    uint32_t zone = (range << __builtin_clz(range)) - 1;
    for (;;) {
      uint64_t m = (uint64_t)next_u32() * range;
      if ((uint32_t)m <= zone) return lo + (uint32_t)(m >> 32);
    }
  }
  // same for usize (64-bit) ranges, exclusive upper bound: gen_range(0..n)
  uint64_t gen_range_u64_exclusive(uint64_t lo, uint64_t hi) {
    uint64_t range = hi - lo;  // (hi-1) - lo + 1
    uint64_t zone = (range << __builtin_clzll(range)) - 1;
    for (;;) {
      unsigned __int128 m = (unsigned __int128)next_u64() * range;
      if ((uint64_t)m <= zone) return lo + (uint64_t)(m >> 64);
    }
  }
};

// ---------------------------------------------------------------------------
// bvh.rs
// ---------------------------------------------------------------------------
template <class R> struct BVHNode : Hittable<R> {
  HPtr<R> left, right;
  Aabb<R> box;
  int level = 0;  // 0: the world's tree (render.rs:211); 1: a tree that is an object of the world (scenes.rs:449,548)
  // bvh.rs:19-76.  `axis_rng` stands in for thread_rng().gen_range(0..=2).
  BVHNode(const std::vector<HPtr<R>>& src, size_t begin, size_t end, R time0, R time1, Pcg32& axis_rng, int tree_level = 0) : level(tree_level) {
    std::vector<HPtr<R>> objs(src.begin() + begin, src.begin() + end);
    int axis = (int)axis_rng.gen_range_u32_inclusive(0, 2);
    auto key = [axis](const HPtr<R>& h) -> R {  // bvh.rs:83-94 (boxes at time (0,0))
      auto b = h->bounding_box(0, 0);
      return b ? b->minimum[axis] : (R)0;
    };
    auto less = [&](const HPtr<R>& a, const HPtr<R>& b) { return total_less(key(a), key(b)); };
    size_t n = objs.size();
    if (n == 1) {
      left = right = objs[0];
    } else if (n == 2) {
      if (less(objs[0], objs[1])) { left = objs[0]; right = objs[1]; }
      else { left = objs[1]; right = objs[0]; }
    } else {
      std::stable_sort(objs.begin(), objs.end(), less);  // slice::sort_by is stable
      size_t mid = n / 2;
      left = std::make_shared<BVHNode>(objs, 0, mid, time0, time1, axis_rng, tree_level);
      right = std::make_shared<BVHNode>(objs, mid, n, time0, time1, axis_rng, tree_level);
    }
    auto bl = left->bounding_box(time0, time1);
    auto br = right->bounding_box(time0, time1);
    Aabb<R> e{};
    box = (bl ? *bl : e).surrounding_box(br ? *br : e);
  }
  static bool total_less(R a, R b) {  // f64::total_cmp(...).is_lt()
    if (std::isnan(a) || std::isnan(b)) return false;  // no NaN boxes occur in any scene
    if (a == b) return std::signbit(a) && !std::signbit(b);
    return a < b;
  }
  bool hit(const Ray<R>& ray, R t_min, R t_max, HitRecord<R>& rec, HitCtx& ctx) const override {  // bvh.rs:108-119
    if (!box.hit(ray, t_min, t_max)) return false;
    bool hit_left = left->hit(ray, t_min, t_max, rec, ctx);
    R right_t_max = hit_left ? rec.t : t_max;
    const uint32_t outer = ctx.visit;
    if (right.get() == left.get()) ctx.visit = outer | (1u << level);  // the same object a second time
    bool hit_right = right->hit(ray, t_min, right_t_max, rec, ctx);
    ctx.visit = outer;
    return hit_left || hit_right;
  }
  std::optional<Aabb<R>> bounding_box(R, R) const override { return box; }  // bvh.rs:121-123
  // Leaves of a nested BVH keep the sub index they were given before the tree was built.
  void label(int obj, int) override {
    left->label(obj, KEEP_SUB);
    if (right.get() != left.get()) right->label(obj, KEEP_SUB);
  }
};

// ---------------------------------------------------------------------------
// camera.rs
// ---------------------------------------------------------------------------
template <class R> struct Camera {
  V3<R> origin, lower_left_corner, horizontal, vertical, u, v, w;
  R lens_radius = 0, shutter_open = 0, shutter_close = 0;
  Camera() = default;
  Camera(V3<R> look_from, V3<R> look_at, V3<R> v_up, R vfov_degrees, R aspect_ratio, R aperture, R focus_dist, R t_open, R t_close) {  // camera.rs:21-60
    R theta = vfov_degrees * (R)(3.14159265358979323846264338327950288 / 180.0);
    R h = std::tan(theta / (R)2);
    R viewport_height = (R)2 * h;
    R viewport_width = aspect_ratio * viewport_height;
    w = unit_vector(look_from - look_at);
    u = unit_vector(v_up.cross(w));
    v = w.cross(u);
    origin = look_from;
    horizontal = (focus_dist * viewport_width) * u;
    vertical = (focus_dist * viewport_height) * v;
    lower_left_corner = origin - horizontal / (R)2 - vertical / (R)2 - focus_dist * w;
    lens_radius = aperture / (R)2;
    shutter_open = t_open;
    shutter_close = t_close;
  }
  Ray<R> get_ray(R s, R t, Stream& rng) const {  // camera.rs:62-72
    V3<R> rd = lens_radius * random_in_unit_disk<R>(rng);
    V3<R> offset = u * rd.x() + v * rd.y();
    V3<R> direction = lower_left_corner + s * horizontal + t * vertical - origin - offset;
    R time = gen_range_at<R>(rng, 2, shutter_open, shutter_close);
    return Ray<R>{origin + offset, direction, time};
  }
};

// ---------------------------------------------------------------------------
// color.rs
// ---------------------------------------------------------------------------
template <class R> struct Background {
  int kind = 0;  // 0 = Solid(c0), 1 = Gradient(start=c0, end=c1)
  V3<R> c0, c1;
  V3<R> color(const V3<R>& direction) const {  // color.rs:42-50
    ++g_cnt.bg;
    if (kind == 0) return c0;
    V3<R> ud = unit_vector(direction);
    R t = (R)0.5 * (ud.y() + (R)1);
    return ((R)1 - t) * c1 + t * c0;
  }
};
inline uint8_t sat_u8(double x) {  // Rust `as u8`
  if (!(x > 0.0)) return 0;
  if (x >= 255.0) return 255;
  return (uint8_t)x;
}
inline void color_to_rgb(const double sum[3], uint32_t spp, uint8_t out[3]) {  // color.rs:14-27
  double scale = 1.0 / (double)spp;
  for (int c = 0; c < 3; ++c) {
    double x = std::sqrt(scale * sum[c]);
    double cl = x;  // f64::clamp(0.0, 0.999): NaN stays NaN
    if (cl < 0.0) cl = 0.0;
    if (cl > 0.999) cl = 0.999;
    out[c] = sat_u8(256.0 * cl);
  }
}

// ---------------------------------------------------------------------------
// render.rs:149-177 ray_color
// ---------------------------------------------------------------------------
// Deliberate faults (mutation checks of the test suite: each must make a known-answer test fail; 0 = the reference's behaviour).
enum Mutation { MUT_NONE = 0, MUT_ONE_SIDED_LIGHT = 1, MUT_DIVIDE_BY_W_H = 2, MUT_TMIN_ZERO = 3 };
inline int g_mutation = MUT_NONE;
struct PathKey {
  uint64_t seed;
  uint32_t pixel, sample;
};
// Optional trace of one path (test instrumentation): every closest-hit query appends its ray and what it found.
struct PathVertex {
  double o[3], d[3], t;
  int obj, sub, hit, front;
};
inline thread_local std::vector<PathVertex>* g_path_trace = nullptr;
template <class R>
V3<R> ray_color(const Ray<R>& ray, const Background<R>& bg, const Hittable<R>& world, uint32_t depth, uint32_t max_depth, const PathKey& k) {
  if (depth == 0) { ++g_cnt.depth_exhausted; return V3<R>(); }
  ++g_cnt.rays;
  uint32_t bounce = max_depth - depth;
  Stream media(k.seed, k.pixel, k.sample, bounce, STAGE_MEDIA);
  HitCtx ctx{&media};
  HitRecord<R> rec;
  const bool was_hit = world.hit(ray, g_mutation == MUT_TMIN_ZERO ? (R)0 : (R)0.001, std::numeric_limits<R>::infinity(), rec, ctx);
  if (g_path_trace) {
    PathVertex v{{(double)ray.origin[0], (double)ray.origin[1], (double)ray.origin[2]},
                 {(double)ray.direction[0], (double)ray.direction[1], (double)ray.direction[2]},
                 was_hit ? (double)rec.t : 0.0, was_hit ? rec.obj : -1, was_hit ? rec.sub : -1, was_hit ? 1 : 0, was_hit && rec.front_face ? 1 : 0};
    g_path_trace->push_back(v);
  }
  if (!was_hit) return bg.color(ray.direction);
  V3<R> emitted = rec.material->emitted(rec.u, rec.v, rec.p);
  if (g_mutation == MUT_ONE_SIDED_LIGHT && !rec.front_face) emitted = V3<R>();
  Stream sc(k.seed, k.pixel, k.sample, bounce, STAGE_SCATTER);
  auto res = rec.material->scatter(ray, rec, sc);
  if (!res) return emitted;
  return emitted + res->attenuation * ray_color(res->scattered, bg, world, depth - 1, max_depth, k);
}

}  // namespace orc
