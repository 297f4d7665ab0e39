// The f64 quotient a / b formed from the exactly rounded reciprocal of b (the kernels have it at hand: 1/d of a ray direction) with
// one correction step - q = a * r, q' = q + (a - b q) r, then the division sequence's own fix-up instruction for zero / infinite /
// NaN operands - compared bit for bit with the compiler's a / b on 2^22 random pairs and a table of special operands.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__device__ double rcp_exact(double a) {
  double r0 = __builtin_amdgcn_rcp(a);
  double r1 = __builtin_fma(__builtin_fma(-a, r0, 1.0), r0, r0);
  double r2 = __builtin_fma(__builtin_fma(-a, r1, 1.0), r1, r1);
  return __builtin_amdgcn_div_fixup(r2, a, 1.0);
}
__global__ void k(const double* a, const double* b, unsigned long long* got, unsigned long long* want, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = a[i], d = b[i], r = rcp_exact(d);
  const double q = x * r;
  const double q1 = __builtin_fma(__builtin_fma(-q, d, x), r, q);
  got[i] = (unsigned long long)__double_as_longlong(__builtin_amdgcn_div_fixup(q1, d, x));
  want[i] = (unsigned long long)__double_as_longlong(x / d);
}

int main() {
  std::vector<double> a = {0.0, 1.0, -1.0, 0.0, -0.0, 1.0, INFINITY, 1.0, NAN, 1.0, 0.0, INFINITY, 3.0, 1e300, 1e-300, 555.0, -277.5, 1.0, 2.0, 1e-320},
                      b = {0.0, 0.0, 0.0, 1.0, 1.0, INFINITY, 1.0, NAN, 1.0, -0.0, -0.0, INFINITY, 3.0, 1e-10, 1e10, 1e-320, 4.9e-324, 1.9999999999999998, 1.9999999999999998, 3.0};
  unsigned long long s = 88172645463325252ull;
  const int n_random = 1 << 22;
  for (int i = 0; i < n_random; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double x = std::ldexp(1.0 + (double)(s >> 12) * 0x1.0p-52, (int)(s % 41) - 20) * ((s & 1) ? -1.0 : 1.0);
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double d = std::ldexp(1.0 + (double)(s >> 12) * 0x1.0p-52, (int)(s % 41) - 20) * ((s & 2) ? -1.0 : 1.0);
    a.push_back(x); b.push_back(d);
  }
  const int n = (int)a.size();
  double *da, *db; unsigned long long *dg, *dw;
  hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dg, n * 8); hipMalloc(&dw, n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dg, dw, n);
  std::vector<unsigned long long> g(n), w(n);
  hipMemcpy(g.data(), dg, n * 8, hipMemcpyDeviceToHost); hipMemcpy(w.data(), dw, n * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const bool both_nan = (g[i] << 1) > 0xffe0000000000000ull && (w[i] << 1) > 0xffe0000000000000ull;
    if (g[i] != w[i] && !both_nan && bad++ < 10) printf("  %.17g / %.17g: corrected %016llx, division %016llx\n", a[i], b[i], g[i], w[i]);
  }
  printf("corrected quotient vs division: %d of %d pairs differ\n", bad, n);
  return bad ? 1 : 0;
}
