// Accuracy of the hardware f64 reciprocal / reciprocal square root on gfx950, raw and after Newton steps, against IEEE division
// and square root (printed in ulps of the result).  Decides how many refinement steps the small-scene f64 kernels use.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = x[i];
  double r0 = __builtin_amdgcn_rcp(a);
  double r1 = __builtin_fma(__builtin_fma(-a, r0, 1.0), r0, r0);
  double r2 = __builtin_fma(__builtin_fma(-a, r1, 1.0), r1, r1);
  double s0 = __builtin_amdgcn_rsq(a);
  // y' = y + y * (0.5 - 0.5 * a * y * y)
  double h0 = 0.5 * s0, e0 = __builtin_fma(-a * s0, h0, 0.5);
  double s1 = __builtin_fma(s0, e0, s0);
  double h1 = 0.5 * s1, e1 = __builtin_fma(-a * s1, h1, 0.5);
  double s2 = __builtin_fma(s1, e1, s1);
  out[i * 8 + 0] = r0; out[i * 8 + 1] = r1; out[i * 8 + 2] = r2; out[i * 8 + 3] = 1.0 / a;
  out[i * 8 + 4] = s0; out[i * 8 + 5] = s1; out[i * 8 + 6] = s2; out[i * 8 + 7] = 1.0 / __builtin_sqrt(a);
}

// The reciprocal as the kernels use it: two Newton steps, then the division sequence's own last instruction for the arguments
// (zero, infinity, NaN, subnormal) where the steps do not apply.  Compared bit for bit with the compiler's 1.0 / x.
__global__ void k_fixup(const double* x, unsigned long long* got, unsigned long long* want, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = x[i];
  double r0 = __builtin_amdgcn_rcp(a);
  double r1 = __builtin_fma(__builtin_fma(-a, r0, 1.0), r0, r0);
  double r2 = __builtin_fma(__builtin_fma(-a, r1, 1.0), r1, r1);
  got[i] = (unsigned long long)__double_as_longlong(__builtin_amdgcn_div_fixup(r2, a, 1.0));
  want[i] = (unsigned long long)__double_as_longlong(1.0 / a);
}

static void check_fixup(const std::vector<double>& random_x) {
  std::vector<double> x = {0.0, -0.0, INFINITY, -INFINITY, NAN, 4.9e-324, -4.9e-324, 1e-310, 2.2250738585072014e-308, 1.7976931348623157e308,
                           -1.7976931348623157e308, 8.98846567431158e307, 4.49423283715579e307, 1.0, -1.0, 3.0, 1e-300, 1e300};
  x.insert(x.end(), random_x.begin(), random_x.end());
  const int n = (int)x.size();
  double* dx; unsigned long long *dg, *dw;
  hipMalloc(&dx, n * 8); hipMalloc(&dg, n * 8); hipMalloc(&dw, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_fixup, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dg, dw, n);
  std::vector<unsigned long long> g(n), w(n);
  hipMemcpy(g.data(), dg, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(w.data(), dw, n * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const bool both_nan = (g[i] << 1) > 0xffe0000000000000ull && (w[i] << 1) > 0xffe0000000000000ull;
    if (g[i] != w[i] && !both_nan) {
      if (bad++ < 10) printf("  1/%.17g: refined %016llx, division %016llx\n", x[i], g[i], w[i]);
    }
  }
  printf("refined reciprocal with fix-up vs division: %d of %d arguments differ\n", bad, n);
}

int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), out((size_t)n * 8);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    x[i] = std::ldexp(1.0 + (double)(s >> 12) * 0x1.0p-52, (int)(s % 41) - 20);
  }
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, (size_t)n * 64);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(out.data(), dout, (size_t)n * 64, hipMemcpyDeviceToHost);
  const char* names[8] = {"rcp raw", "rcp + 1 step", "rcp + 2 steps", "", "rsq raw", "rsq + 1 step", "rsq + 2 steps", ""};
  for (int k = 0; k < 8; ++k) {
    if (k == 3 || k == 7) continue;
    double worst = 0, sum = 0;
    const int ref = k < 3 ? 3 : 7;
    for (int i = 0; i < n; ++i) {
      double want = k < 3 ? 1.0 / x[i] : 1.0 / std::sqrt(x[i]);
      (void)ref;
      double ulp = std::ldexp(1.0, std::ilogb(want) - 52);
      double e = std::fabs(out[(size_t)i * 8 + k] - want) / ulp;
      worst = e > worst ? e : worst; sum += e;
    }
    printf("%-14s max %.3g ulp, mean %.3g ulp\n", names[k], worst, sum / n);
  }
  check_fixup(x);
  return 0;
}
