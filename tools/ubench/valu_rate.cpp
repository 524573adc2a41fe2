// Micro-benchmark (diagnostic, not product): VALU / v_exp_f32 issue rates on gfx950 vs waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rate tools/ubench/valu_rate.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITERS = 4096;

// MODE 0: 16 fma per iter (8 independent chains x2); MODE 1: 8 exp + 8 fma; MODE 2: scan step mix x4 states
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float x[8], a = seed + threadIdx.x * 1e-7f, b = 0.999f;
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = seed * j;
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = fmaf(x[j], b, a);
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = fmaf(x[j], b, a);
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = fmaf(__builtin_amdgcn_exp2f(x[j]), b, a);
    } else {
      // 2 steps x 4 states: mul, exp, mul, fma, fma  (= 5 VALU incl. 1 exp per state-step)
      float y = 0.f;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float dl = a + e * 1e-3f + y * 1e-9f, du = b + e;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float aa = __builtin_amdgcn_exp2f(dl * (-1.f - j));
          x[j] = fmaf(aa, x[j], du * (0.5f + j));
          y = fmaf(x[j], 0.25f + j, y);
        }
      }
      x[7] += y;
    }
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int valu_per_iter, int exp_per_iter, float* d) {
  for (int wps = 1; wps <= 6; ++wps) {
    const int blocks = 256 * wps;   // 256-thread blocks: 4 waves -> one per SIMD; wps blocks per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE><<<blocks, 256>>>(d, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    // per SIMD: wps waves x ITERS x instr
    const double instr = (double)wps * ITERS * (valu_per_iter + exp_per_iter);
    printf("%-10s waves/SIMD %d  %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)  [valu %d exp %d per iter]\n",
           name, wps, ms, ms * 1e6 / instr, ms * 1e6 / instr * 2.4, valu_per_iter, exp_per_iter);
  }
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<0>("fma", 16, 0, d);
  run<1>("exp+fma", 8, 8, d);
  run<2>("scanmix", 32 + 2, 8, d);
  return 0;
}
