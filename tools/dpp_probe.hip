// DPP throughput probe: what do the cross-lane moves of the wave kernels cost next to v_fma_f64?
//   hipcc --offload-arch=gfx950 -O3 tools/dpp_probe.hip -o tools/bin/dpp_probe && tools/bin/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
// MODE 0: v_mov_b32 (no DPP)  1: row_ror:1 (inside 16 lanes)  2: wave_ror:1 (all 64 lanes)  3: wave_ror:1 + fma mix
template <int MODE>
__global__ void __launch_bounds__(64) k(int* out, int iters, double a, double b) {
  int x[8];
  double f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    x[i] = threadIdx.x * 8 + i;
    f[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) x[i] = x[i] ^ 0x5a5a;  // a plain 32-bit VALU op per register
      if (MODE == 1) x[i] = __builtin_amdgcn_mov_dpp(x[i], 0x121, 0xF, 0xF, false);
      if (MODE == 2 || MODE == 3) x[i] = __builtin_amdgcn_mov_dpp(x[i], 0x13C, 0xF, 0xF, false);
      if (MODE == 3) {
        f[i] = fma(f[i], a, b);
        f[i] = fma(f[i], a, b);
      }
    }
  }
  int s = 0;
  double t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s += x[i];
    t += f[i];
  }
  out[blockIdx.x * 64 + threadIdx.x] = s + (int)t;
}
template <int MODE>
void run(const char* name, int iters) {
  const int blocks = 256 * 4 * 2;
  int* d;
  (void)hipMalloc(&d, sizeof(int) * blocks * 64);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d, iters, 0.9999999, 1e-12);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 waves x 8 ops per iteration
    printf("%-34s %.3f ms  %.2f ns per wave-instruction and SIMD (%.1f cycles at 2.33 GHz)%s\n", name, ms,
           ms * 1e6 / (2.0 * 8 * iters), ms * 1e6 / (2.0 * 8 * iters) * 2.33, MODE == 3 ? "  [per mov + 2 fma]" : "");
  }
  (void)hipFree(d);
}
int main() {
  run<0>("v_xor_b32", 200000);
  run<1>("v_mov_b32_dpp row_ror:1", 200000);
  run<2>("v_mov_b32_dpp wave_ror:1", 200000);
  run<3>("wave_ror:1 + 2 v_fma_f64", 100000);
  return 0;
}
