// Seeded generator used wherever the reference's scene construction draws random numbers.
//
// PCG-XSH-RR 64/32 (O'Neill, 2014) with the seeding of rand_core 0.6.4 `SeedableRng::seed_from_u64`
// and rand_pcg 0.3.1 `Lcg64Xsh32::from_seed`, which is what `rand_pcg::Pcg32::seed_from_u64(2)` and
// `(3)` resolve to in the reference (src/scenes.rs:158,439).  Value conversions follow rand 0.8.5:
// `gen::<f64>()` keeps the top 53 bits, `gen_range(lo..hi)` builds a [1,2) mantissa, integer ranges use
// widening-multiply rejection.  Those crates are not part of the reference tree; this is a restatement
// of their published algorithms and cannot be compared with a Rust run offline.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>

namespace rth {

extern thread_local std::string g_err;

class Pcg32 {
 public:
  static Pcg32 seed_from_u64(uint64_t seed) {
    uint32_t words[4];
    for (uint32_t& w : words) {
      seed = seed * kMul + 11634580027462260723ull;
      w = rotr((uint32_t)(((seed >> 18) ^ seed) >> 27), (uint32_t)(seed >> 59));
    }
    Pcg32 g;
    g.inc_ = (((uint64_t)words[3] << 32) | words[2]) | 1ull;
    g.state_ = (((uint64_t)words[1] << 32) | words[0]) + g.inc_;
    g.state_ = g.state_ * kMul + g.inc_;
    return g;
  }
  uint32_t u32() {
    uint64_t old = state_;
    state_ = old * kMul + inc_;
    return rotr((uint32_t)(((old >> 18) ^ old) >> 27), (uint32_t)(old >> 59));
  }
  uint64_t u64() {
    uint64_t lo = u32();
    uint64_t hi = u32();
    return (hi << 32) | lo;
  }
  double f64() { return (double)(u64() >> 11) * 0x1.0p-53; }
  double range_f64(double lo, double hi) {
    double span = hi - lo;
    for (;;) {
      double v = ((double)(u64() >> 12) * 0x1.0p-52) * span + lo;
      if (v < hi) return v;
      span = std::nextafter(span, 0.0);
    }
  }
  uint32_t range_u32_inclusive(uint32_t lo, uint32_t hi) {
    uint32_t n = hi - lo + 1u;
    if (n == 0) return u32();
    uint32_t zone = (n << __builtin_clz(n)) - 1u;
    for (;;) {
      uint64_t wide = (uint64_t)u32() * n;
      if ((uint32_t)wide <= zone) return lo + (uint32_t)(wide >> 32);
    }
  }
  // uniform in [0, n), n >= 1, 64-bit (usize) flavour
  uint64_t below_u64(uint64_t n) {
    uint64_t zone = (n << __builtin_clzll(n)) - 1ull;
    for (;;) {
      unsigned __int128 wide = (unsigned __int128)u64() * n;
      if ((uint64_t)wide <= zone) return (uint64_t)(wide >> 64);
    }
  }

 private:
  static constexpr uint64_t kMul = 6364136223846793005ull;
  static uint32_t rotr(uint32_t v, uint32_t r) { return (v >> r) | (v << ((32u - r) & 31u)); }
  uint64_t state_ = 0, inc_ = 1;
};

}  // namespace rth
