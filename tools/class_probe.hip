// Probe for the genotype-class sweep: acc[k] *= X[c_k] for k = 0..63 with a wave-uniform class c_k in {0..3} taken from
// 128 scalar bits per entry, X[0..3] per lane.  Which way of selecting X by a uniform index is cheapest on gfx950?
//   MODE 0  reference: acc[k] *= X[k & 3] (static index: the floor, one v_mul_f64 per step)
//   MODE 1  switch on the scalar class (uniform branches)
//   MODE 2  X[c] as a register array indexed by the uniform class (VGPR index mode)
//   MODE 3  X staged in LDS, ds_read_b64 at a uniform offset
//   MODE 4  branch-free cndmask tree on the scalar class
//   MODE 5  VGPR index mode (s_set_gpr_idx_on / _idx, src0 relative) in inline assembly, sixteen steps per statement
//   hipcc --offload-arch=gfx950 -O3 tools/class_probe.hip -o tools/bin/class_probe && tools/bin/class_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(64, 2) k(const uint32_t* __restrict__ bits, double* out, int iters, double a) {
  __shared__ double lx[4][64];
  double acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = 1.0;
  const int lane = threadIdx.x;
  double X[4];
  for (int it = 0; it < iters; ++it) {
    const uint32_t* w = bits + (size_t)((blockIdx.x * 131 + it) & 4095) * 4;  // wave-uniform address: scalar loads
    const uint32_t w0 = __builtin_amdgcn_readfirstlane(w[0]), w1 = __builtin_amdgcn_readfirstlane(w[1]),
                   w2 = __builtin_amdgcn_readfirstlane(w[2]), w3 = __builtin_amdgcn_readfirstlane(w[3]);
    const uint32_t ww[4] = {w0, w1, w2, w3};
#pragma unroll
    for (int b = 0; b < 4; ++b) X[b] = 1.0 + a * (double)(lane + b + (it & 7));
    if (MODE == 3) {
#pragma unroll
      for (int b = 0; b < 4; ++b) lx[b][lane] = X[b];
    }
    if (MODE == 5) {
      typedef double d4 __attribute__((ext_vector_type(4)));
      d4 XV = {X[0], X[1], X[2], X[3]};
#define STEP(i) "s_bfe_u32 %16, %17, " #i "*2|0x20000\n\ts_lshl_b32 %16, %16, 1\n\ts_set_gpr_idx_idx %16\n\tv_mul_f64 %" #i ", v[200:201], %" #i "\n\t"
#define GROUP(g, W)                                                                                                      \
  {                                                                                                                      \
    uint32_t tmp;                                                                                                        \
    asm volatile("s_mov_b32 %16, 0\n\ts_set_gpr_idx_on %16, 0x1\n\t" STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) \
                     STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15) "s_set_gpr_idx_off"   \
                 : "+v"(acc[g * 16 + 0]), "+v"(acc[g * 16 + 1]), "+v"(acc[g * 16 + 2]), "+v"(acc[g * 16 + 3]),          \
                   "+v"(acc[g * 16 + 4]), "+v"(acc[g * 16 + 5]), "+v"(acc[g * 16 + 6]), "+v"(acc[g * 16 + 7]),          \
                   "+v"(acc[g * 16 + 8]), "+v"(acc[g * 16 + 9]), "+v"(acc[g * 16 + 10]), "+v"(acc[g * 16 + 11]),        \
                   "+v"(acc[g * 16 + 12]), "+v"(acc[g * 16 + 13]), "+v"(acc[g * 16 + 14]), "+v"(acc[g * 16 + 15]),      \
                   "=&s"(tmp)                                                                                            \
                 : "s"(W), "{v[200:207]}"(XV));                                                                                                \
  }
      GROUP(0, w0) GROUP(1, w1) GROUP(2, w2) GROUP(3, w3)
    }
#pragma unroll
    for (int kk = 0; kk < 64 && MODE != 5; ++kk) {
      const uint32_t c = (ww[kk >> 4] >> (2 * (kk & 15))) & 3u;
      if (MODE == 0) {
        acc[kk] *= X[kk & 3];
      } else if (MODE == 1) {
        switch (c) {
          case 0: acc[kk] *= X[0]; break;
          case 1: acc[kk] *= X[1]; break;
          case 2: acc[kk] *= X[2]; break;
          default: acc[kk] *= X[3]; break;
        }
      } else if (MODE == 2) {
        acc[kk] *= X[c];
      } else if (MODE == 3) {
        acc[kk] *= lx[c][lane];
      } else {
        const double lo = (c & 1u) ? X[1] : X[0], hi = (c & 1u) ? X[3] : X[2];
        acc[kk] *= (c & 2u) ? hi : lo;
      }
    }
    if ((it & 15) == 15) {
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        int e;
        acc[i] = frexp(acc[i], &e);
      }
    }
  }
  double t = 0;
#pragma unroll
  for (int i = 0; i < 64; ++i) t += acc[i];
  out[blockIdx.x * 64 + lane] = t;
}

static std::vector<double> g_ref;
template <int MODE>
void run(const char* name, int iters, const uint32_t* d_bits) {
  const int blocks = 256 * 4 * 2;
  double* d;
  (void)hipMalloc(&d, sizeof(double) * blocks * 64);
  {  // same inputs, fewer iterations: the selections of every mode must give the same products as MODE 2
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d_bits, d, 37, 1e-3);
    std::vector<double> h(blocks * 64);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    if (MODE == 2) g_ref = h;
    if (MODE > 2 || MODE == 1) {
      size_t bad = 0;
      for (size_t i = 0; i < h.size(); ++i) bad += h[i] != g_ref[i];
      printf("%-44s %zu of %zu outputs differ from the register-array variant\n", name, bad, h.size());
    }
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), 0, 0, d_bits, d, iters, 1e-9);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 waves x 64 steps per iteration
    const double ns = ms * 1e6 / (2.0 * 64 * iters);
    printf("%-44s %.3f ms  %.2f ns per step and SIMD (%.1f cycles at 2.4 GHz; v_mul_f64 alone = 4)\n", name, ms, ns, ns * 2.4);
  }
  (void)hipFree(d);
}

int main() {
  std::vector<uint32_t> bits(4096 * 4);
  uint32_t s = 12345;
  for (auto& b : bits) {
    s = s * 1664525u + 1013904223u;
    b = s;
  }
  uint32_t* d_bits;
  (void)hipMalloc(&d_bits, bits.size() * 4);
  (void)hipMemcpy(d_bits, bits.data(), bits.size() * 4, hipMemcpyHostToDevice);
  run<2>("register array, uniform index", 20000, d_bits);
  run<0>("static index (floor)", 20000, d_bits);
  run<1>("switch on the uniform class", 20000, d_bits);
  run<3>("LDS table, uniform offset", 20000, d_bits);
  run<4>("cndmask tree", 20000, d_bits);
  run<5>("VGPR index mode (inline asm)", 20000, d_bits);
  return 0;
}
