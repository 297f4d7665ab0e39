// Issue-rate microbenchmark for the instruction classes the bounce kernel is made of (gfx950).
// Every workgroup runs ITER iterations of an unrolled chain of independent instructions; the host reports
// cycles per wave64 instruction per SIMD at a given number of resident waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 4096;
constexpr int CHAINS = 8;

template <int KIND> __global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
  uint32_t a[CHAINS];
  float f[CHAINS];
  double d[CHAINS];
  for (int i = 0; i < CHAINS; ++i) { a[i] = seed + threadIdx.x * 17 + i; f[i] = 1.0f + (float)(a[i] & 255) * 1e-3f; d[i] = f[i]; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      if constexpr (KIND == 0) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);
      if constexpr (KIND == 1) { uint64_t p = (uint64_t)a[i] * 0xD2511F53u + it; a[i] = (uint32_t)p ^ (uint32_t)(p >> 32); }
      if constexpr (KIND == 2) a[i] = a[i] * 0xD2511F53u + 1u;
      if constexpr (KIND == 3) f[i] = __builtin_amdgcn_rcpf(f[i]) + 1.0f;
      if constexpr (KIND == 4) d[i] = __builtin_fma(d[i], 1.0001, 0.5);
      if constexpr (KIND == 5) d[i] = 1.0 / d[i] + 1.0;
      if constexpr (KIND == 6) f[i] = 1.0f / f[i] + 1.0f;
      if constexpr (KIND == 7) a[i] = (a[i] ^ (a[i] >> 7)) + 0x9E3779B9u;
      if constexpr (KIND == 8) f[i] = __builtin_sqrtf(f[i]) + 1.0f;
      if constexpr (KIND == 9) d[i] = __builtin_sqrt(d[i]) + 1.0;
      if constexpr (KIND == 10) a[i] = __builtin_amdgcn_ds_bpermute((int)(a[i] & 252), (int)a[i]) + 1;
      if constexpr (KIND == 11) f[i] = __builtin_amdgcn_sqrtf(f[i]) + 1.0f;
      if constexpr (KIND == 12) d[i] = (double)a[i] + d[i];
      if constexpr (KIND == 13) d[i] = __builtin_ldexp(d[i], (int)(a[i] & 1u));
      if constexpr (KIND == 14) { uint64_t x = (((uint64_t)a[i] << 31) | (uint64_t)(uint32_t)it) >> 11; a[i] = (uint32_t)x ^ (uint32_t)(x >> 32); }
      if constexpr (KIND == 15) d[i] = (double)((((uint64_t)a[i]) << 20) | (uint64_t)(uint32_t)it) + d[i];
      if constexpr (KIND == 16) d[i] = d[i] > 1.5 ? d[i] - 0.5 : d[i] + 0.25;
      if constexpr (KIND == 17) d[i] = __builtin_fmax(d[i], 1.0001) + 0.001;
      if constexpr (KIND == 18) d[i] = d[i] + 1.0001;
      if constexpr (KIND == 19) d[i] = d[i] * 1.0001;
      if constexpr (KIND == 20) a[i] = (a[i] & 1u) ? a[i] + 3u : a[i] ^ 5u;
      if constexpr (KIND == 21) f[i] = (float)a[i] + f[i];
      if constexpr (KIND == 22) a[i] += (uint32_t)__builtin_amdgcn_readlane((int)a[i], 3);
      if constexpr (KIND == 23) f[i] = f[i] > 1.5f ? f[i] - 0.5f : f[i] + 0.25f;
      if constexpr (KIND == 24) { float lo = 1.0f, hi = 2.0f; a[i] += (__builtin_amdgcn_fmed3f(f[i], lo, hi) == f[i]) ? 1u : 0u; f[i] += 0.001f; }
      if constexpr (KIND == 26) a[i] = __umulhi(a[i], 0xD2511F53u) + (uint32_t)it;
      if constexpr (KIND == 27) { uint32_t hi = __umulhi(a[i], 0xD2511F53u), lo = a[i] * 0xD2511F53u; a[i] = hi ^ lo ^ (uint32_t)it; }
      if constexpr (KIND == 25) { a[i] += ((d[i] >= 1.0) & (d[i] <= 2.0)) ? 1u : 0u; d[i] += 0.001; }
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < CHAINS; ++i) r ^= a[i] ^ __float_as_uint(f[i]) ^ (uint32_t)__double_as_longlong(d[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND> int run(const char* name, int instr_per_op, uint32_t* d_out) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int waves_per_simd : {1, 2, 4, 8}) {
    int blocks = 256 * waves_per_simd;  // 256 CUs, 4 waves per block -> one wave per SIMD per block
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 2u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ops_per_simd = (double)ITER * CHAINS * waves_per_simd;
    double cyc = ms * 1e-3 * 2.4e9 / ops_per_simd;
    printf("%-28s waves/SIMD %d: %8.3f ms  %6.2f cyc per op per SIMD at 2.4 GHz (%d instr/op nominal)\n", name, waves_per_simd, ms, cyc, instr_per_op);
  }
  return 0;
}

int main() {
  uint32_t* d_out;
  CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
  run<0>("v_fma_f32", 1, d_out);
  run<7>("xor/shift/add u32", 3, d_out);
  run<1>("mad_u64_u32 + xor (philox)", 2, d_out);
  run<2>("mul_lo_u32 + add", 1, d_out);
  run<3>("v_rcp_f32 + add", 2, d_out);
  run<6>("1.0f/x IEEE + add", 10, d_out);
  run<8>("sqrtf IEEE + add", 10, d_out);
  run<11>("v_sqrt_f32 + add", 2, d_out);
  run<4>("v_fma_f64", 1, d_out);
  run<5>("1.0/x f64 + add", 12, d_out);
  run<9>("sqrt f64 + add", 12, d_out);
  run<10>("ds_bpermute + add", 2, d_out);
  run<26>("v_mul_hi_u32 + add", 2, d_out);
  run<27>("mul_hi + mul_lo + 2 xor", 4, d_out);
  run<18>("v_add_f64", 1, d_out);
  run<19>("v_mul_f64", 1, d_out);
  run<12>("v_cvt_f64_u32 + add_f64", 2, d_out);
  run<13>("v_ldexp_f64", 1, d_out);
  run<14>("u64 shift/or + xor", 4, d_out);
  run<15>("u64 -> f64 + add_f64", 6, d_out);
  run<16>("f64: cmp + 2 arith + 2 cndmask", 5, d_out);
  run<17>("v_max_f64 + add_f64", 2, d_out);
  run<25>("f64: 2 cmp + and + addc, add_f64", 5, d_out);
  run<20>("u32: and/cmp + add + xor + cndmask", 5, d_out);
  run<21>("v_cvt_f32_u32 + add_f32", 2, d_out);
  run<22>("v_readlane + add", 2, d_out);
  run<23>("f32: cmp + 2 arith + cndmask", 4, d_out);
  run<24>("f32: med3 + cmp + addc, add_f32", 4, d_out);
  CHECK(hipFree(d_out));
  return 0;
}
