// accuracy of pos_log (popscle_amd/csrc/common.hpp) against the host's log: hipcc --offload-arch=gfx950 -O3 -I include
// tools/pos_log_probe.hip -o /tmp/pos_log_probe && /tmp/pos_log_probe
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "../popscle_amd/csrc/common.hpp"

__global__ void probe(int64_t n, const double* x, const int32_t* e, double* o) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = prodacc_log(x[i], e[i]);
}

int main() {
  const int64_t n = 1 << 24;
  std::vector<double> x(n), o(n);
  std::vector<int32_t> e(n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> u(0.0, 1.0);
  for (int64_t i = 0; i < n; ++i) {
    const int kind = (int)(i & 3);
    // mantissas in [0.5, 1), values next to 1 and to sqrt(1/2), products that were not renormalised (down to 2^-900)
    double v = kind == 0 ? 0.5 + 0.5 * u(g) : kind == 1 ? 1.0 - 1e-9 * u(g) : kind == 2 ? 0.70710678118654752 * (1 + 1e-12 * (u(g) - 0.5))
                                                                              : ldexp(0.5 + 0.5 * u(g), -(int)(900 * u(g)));
    x[i] = v;
    e[i] = (int32_t)((u(g) - 0.7) * 400000);
  }
  x[0] = 0.0;
  double *dx, *dout;
  int32_t* de;
  hipMalloc(&dx, n * 8), hipMalloc(&dout, n * 8), hipMalloc(&de, n * 4);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice), hipMemcpy(de, e.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, dx, de, dout);
  hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost);
  double worst_abs = 0, worst_rel = 0;
  for (int64_t i = 1; i < n; ++i) {
    const long double want = logl((long double)x[i]) + (long double)e[i] * 0.693147180559945309417232121458L;
    const double d = (double)fabsl((long double)o[i] - want);
    worst_abs = std::fmax(worst_abs, d);
    worst_rel = std::fmax(worst_rel, d / std::fmax(1.0, (double)fabsl(want)));
  }
  printf("pos_log: %lld values, max |error| %.3e, max error / max(1, |log|) %.3e, log(0) = %g\n", (long long)n, worst_abs, worst_rel, o[0]);
  return !(worst_rel < 4e-16 && std::isinf(o[0]) && o[0] < 0);
}
