// FP64 dependent-issue probe: what does an fma -> mul pair on ONE temporary (the order hipcc emits for
// `acc[i] *= fma(B, P[i], X)` under register pressure) cost against the same instructions with the dependent pair
// D instructions apart, at 1 / 2 / 4 waves per SIMD?  Also a DPP mov feeding an fma.
// NB (round 5): hipcc puts an s_nop between two asm statements when the second reads a register the first wrote, so the
// variants below that are written as separate statements also measure those; tools/fp64_ilp_probe.hip keeps every
// variant's instructions in ONE asm block and is the probe DESIGN.md quotes.
//   hipcc --offload-arch=gfx950 -O3 tools/fp64_dep_probe.hip -o tools/bin/fp64_dep_probe && tools/bin/fp64_dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>

// 16 accumulators; per iteration 16 fma + 16 mul.  DIST = 0: fma_i, mul_i, fma_i+1, mul_i+1 ... (one temp);
// DIST = k: fma_i is issued k pairs ahead of mul_i (k temps).
template <int DIST>
__global__ void __launch_bounds__(64) k(double* out, int iters, double b, double x) {
  double acc[16], p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc[i] = 1.0 + 1e-9 * threadIdx.x;
    p[i] = 1e-9 * (i + threadIdx.x);
  }
  for (int it = 0; it < iters; ++it) {
    if constexpr (DIST == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        double t;
        asm volatile("v_fma_f64 %0, %2, %3, %4\n\tv_mul_f64 %1, %0, %1" : "=&v"(t), "+v"(acc[i]) : "v"(b), "v"(p[i]), "v"(x));
      }
    } else {
      double t[DIST];
#pragma unroll
      for (int i = 0; i < DIST; ++i) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(t[i]) : "v"(b), "v"(p[i]), "v"(x));
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("v_mul_f64 %0, %1, %0" : "+v"(acc[i]) : "v"(t[i % DIST]));
        if (i + DIST < 16) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(t[i % DIST]) : "v"(b), "v"(p[i + DIST]), "v"(x));
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

// a DPP move feeding an fma (mov lo, mov hi, fma) against plain fmas
template <int MODE>
__global__ void __launch_bounds__(64) kd(double* out, int iters, double b, double x) {
  double acc[16], p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc[i] = 1.0 + 1e-9 * threadIdx.x;
    p[i] = 1e-9 * (i + threadIdx.x);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) {  // mov_dpp x2 then the fma that reads them, back to back
        const int lo = __double2loint(p[i]), hi = __double2hiint(p[i]);
        int qlo, qhi;
        asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "=v"(qlo) : "v"(lo));
        asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "=v"(qhi) : "v"(hi));
        const double q = __hiloint2double(qhi, qlo);
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(q), "v"(b));
      } else {
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(p[i]), "v"(b));
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

// the general entries' hypothesis: t = r0 * u0; t = fma(r1, u1, t); t = fma(r2, u2, t); acc *= t -- a chain of four on one
// temporary (IL = 1, the order hipcc emits under register pressure) against IL hypotheses advanced in lock step (IL temps)
template <int IL>
__global__ void __launch_bounds__(64) kg(double* out, int iters, double u0, double u1, double u2) {
  double acc[16], r[3];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 1.0 + 1e-9 * threadIdx.x;
  for (int i = 0; i < 3; ++i) r[i] = 0.3 + 1e-9 * (i + threadIdx.x);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; i += IL) {
      double t[IL];
#pragma unroll
      for (int q = 0; q < IL; ++q) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t[q]) : "v"(r[0]), "v"(u0));
#pragma unroll
      for (int q = 0; q < IL; ++q) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(t[q]) : "v"(r[1]), "v"(u1));
#pragma unroll
      for (int q = 0; q < IL; ++q) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(t[q]) : "v"(r[2]), "v"(u2));
#pragma unroll
      for (int q = 0; q < IL; ++q) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(acc[i + q]) : "v"(t[q]));
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <class F>
void time_it(const char* name, int wps, int iters, double instr_per_iter, F launch) {
  const int blocks = 256 * 4 * wps;
  double* d;
  (void)hipMalloc(&d, sizeof(double) * blocks * 64);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    launch(blocks, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  const double ns = best * 1e6 / (wps * instr_per_iter * iters);  // per wave-instruction and SIMD
  printf("%-44s %d waves/SIMD  %.3f ms  %.2f ns per instruction and SIMD (%.1f cycles at 2.4 GHz)\n", name, wps, best, ns, ns * 2.4);
  (void)hipFree(d);
}

// occupancy is set by the number of resident workgroups: one wave per workgroup, blocks = SIMDs x waves/SIMD; a
// dynamic LDS request keeps more from becoming resident (160 KB per CU / (4 x wps) workgroups)
template <int DIST>
void run(const char* name, int wps) {
  const int lds = 160 * 1024 / (4 * wps) - 1024;
  time_it(name, wps, 40000 / wps, 32.0, [&](int blocks, double* d, int iters) {
    hipLaunchKernelGGL((k<DIST>), dim3(blocks), dim3(64), lds, 0, d, iters, 1e-3, 1.0);
  });
}
template <int IL>
void rung(const char* name, int wps) {
  const int lds = 160 * 1024 / (4 * wps) - 1024;
  time_it(name, wps, 40000 / wps, 64.0, [&](int blocks, double* d, int iters) {
    hipLaunchKernelGGL((kg<IL>), dim3(blocks), dim3(64), lds, 0, d, iters, 0.9, 0.8, 0.7);
  });
}
template <int MODE>
void rund(const char* name, int wps) {
  const int lds = 160 * 1024 / (4 * wps) - 1024;
  time_it(name, wps, 40000 / wps, MODE == 0 ? 48.0 : 16.0, [&](int blocks, double* d, int iters) {
    hipLaunchKernelGGL((kd<MODE>), dim3(blocks), dim3(64), lds, 0, d, iters, 1e-3, 1.0);
  });
}

int main() {
  for (int wps : {1, 2, 4, 8}) {
    run<0>("fma->mul back to back (one temp)", wps);
    run<1>("fma one pair ahead", wps);
    run<2>("fma two pairs ahead", wps);
    run<4>("fma four pairs ahead", wps);
    run<8>("fma eight pairs ahead", wps);
    rung<1>("mul-fma-fma-mul chain, one at a time", wps);
    rung<2>("mul-fma-fma-mul, two in lock step", wps);
    rung<4>("mul-fma-fma-mul, four in lock step", wps);
    rung<8>("mul-fma-fma-mul, eight in lock step", wps);
    rund<1>("independent fma stream (16 chains)", wps);
    rund<0>("2 x mov_dpp -> fma back to back", wps);
  }
  return 0;
}
