// FP64 MFMA probe: throughput of v_mfma_f64_16x16x4_f64 on this box, alone and next to an independent v_mul_f64
// stream in the same wave -- does the matrix pipe take FMA work off the vector ALU for the K = 3 pair sweep?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

// NM independent MFMA accumulators, NV independent multiply chains per wave and iteration
template <int NM, int NV>
__global__ void __launch_bounds__(64) k(double* out, int iters, double a, double b) {
  d4 acc[NM > 0 ? NM : 1];
  double x[NV > 0 ? NV : 1];
#pragma unroll
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = d4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) x[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  const double av = a + 1e-12 * threadIdx.x, bv = b + 1e-12 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i] = x[i] * a;
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NM, int NV>
void run(const char* name, int waves_per_simd, int iters) {
  const int blocks = 256 * 4 * waves_per_simd;
  double* d;
  hipMalloc(&d, sizeof(double) * blocks * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV>), dim3(blocks), dim3(64), 0, 0, d, iters, 0.9999999, 1e-3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)blocks * NM * iters, vm = (double)blocks * NV * iters;
    // per SIMD: waves_per_simd waves, each NM MFMAs per iteration
    const double ns_per_mfma_simd = NM ? ms * 1e6 / ((double)waves_per_simd * NM * iters) : 0;
    printf("%-44s w/SIMD %d  %.3f ms  MFMA %.2f TFLOP/s (%.1f ns per MFMA and SIMD)  VALU mul %.2f T lane-ops/s\n", name,
           waves_per_simd, ms, mf * 2048 / ms * 1e-9, ns_per_mfma_simd, vm * 64 / ms * 1e-9);
  }
  hipFree(d);
}
int main() {
  run<8, 0>("mfma only, 8 accumulators", 2, 20000);
  run<8, 0>("mfma only, 8 accumulators", 4, 10000);
  run<0, 32>("mul only, 32 chains", 2, 20000);
  run<8, 8>("8 mfma + 8 mul per iteration", 2, 20000);
  run<8, 32>("8 mfma + 32 mul per iteration", 2, 20000);
  run<4, 16>("4 mfma + 16 mul per iteration", 2, 20000);
  return 0;
}
