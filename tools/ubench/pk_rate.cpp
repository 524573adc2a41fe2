// Micro-benchmark (diagnostic, not product): v_pk_fma_f32 / v_pk_mul_f32 issue rate on gfx950 vs waves per SIMD,
// next to plain v_fma_f32.  build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/pk_rate tools/ubench/pk_rate.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  v2f x[8];
  v2f a = {seed + threadIdx.x * 1e-7f, seed}, b = {0.999f, 0.998f};
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = (v2f){seed * j, seed};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (MODE == 0) {   // 8 packed fma
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(b), "v"(a));
        } else if constexpr (MODE == 1) {                     // 16 scalar fma (same flops)
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j].x) : "v"(b.x), "v"(a.x));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j].y) : "v"(b.y), "v"(a.y));
        } else {   // packed mul with broadcast of the low half (op_sel_hi:[1,0])
          asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(x[j]) : "v"(b));
        }
      }
  }
  v2f s = {0, 0};
#pragma unroll
  for (int j = 0; j < 8; ++j) s += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <int MODE>
void run(const char* name, int instr_per_iter, int flops_per_lane_iter, float* d) {
  for (int wps = 1; wps <= 4; ++wps) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE><<<blocks, 256>>>(d, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double instr = (double)wps * ITERS * instr_per_iter;
    printf("%-8s waves/SIMD %d  %.3f ms  %.2f cyc per wave-instr per SIMD @2.4GHz   %.1f TFLOP/s\n", name, wps, ms,
           ms * 1e6 / instr * 2.4, (double)blocks * 256 * ITERS * flops_per_lane_iter / ms / 1e9);
  }
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<0>("pk_fma", 16, 64, d);
  run<1>("fma", 32, 64, d);
  run<2>("pk_mul", 16, 32, d);
  return 0;
}
