// FP64 vector peak probe: how close does a pure v_fma_f64 / v_mul_f64 stream get to 78.6 TFLOP/s on this box, with 2 or
// 8 waves per SIMD?  (Answers whether the sweep kernel's FP64 floor is a clock/power limit or a scheduling one.)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS, bool MUL>
__global__ void __launch_bounds__(64) k(double* out, int iters, double a, double b) {
  double x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) x[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) x[i] = MUL ? x[i] * a : fma(x[i], a, b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CHAINS, bool MUL>
void run(const char* name, int blocks, int iters) {
  double* d;
  hipMalloc(&d, sizeof(double) * blocks * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS, MUL>), dim3(blocks), dim3(64), 0, 0, d, iters, 0.9999999, 1e-12);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 64 * CHAINS * iters;
    printf("%-28s blocks %6d  %.3f ms  %.1f G instr-lanes/s  = %.1f TFLOP/s (%s)\n", name, blocks, ms, ops / ms * 1e-6,
           ops * (MUL ? 1 : 2) / ms * 1e-9, MUL ? "mul" : "fma");
  }
  hipFree(d);
}
int main() {
  run<32, false>("fma 32 chains, 8 waves/SIMD", 256 * 4 * 8, 20000);
  run<32, false>("fma 32 chains, 2 waves/SIMD", 256 * 4 * 2, 80000);
  run<32, true>("mul 32 chains, 2 waves/SIMD", 256 * 4 * 2, 80000);
  run<4, false>("fma 4 chains, 2 waves/SIMD", 256 * 4 * 2, 320000);
  return 0;
}
