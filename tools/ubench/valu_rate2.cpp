// Diagnostic micro-benchmark: what breaks the 2-cycle VALU issue rate?  scan-like mix + DPP / LDS traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 2048;
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// MODE bits: 1 = DPP reduce (3 levels) per step, 2 = LDS b128 operand reads (6 per 4 steps), 4 = LDS b128 write per 4 steps
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[4][64 * 36];
  float* my = lds[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  for (int i = lane; i < 64 * 36; i += 64) my[i] = seed + i * 1e-6f;
  float x[2] = {seed, seed * 2}, acc = 0.f;
  const float A2[2] = {-1.f - lane * 1e-3f, -2.f};
  for (int it = 0; it < ITERS; ++it) {
    float4 dl4, du4, B0, B1, C0, C1;
    if constexpr (MODE & 2) {
      const float* base = my + (lane >> 3) * 36 + ((it & 7) << 2);
      dl4 = *reinterpret_cast<const float4*>(base);
      du4 = *reinterpret_cast<const float4*>(base + 8 * 36);
      B0 = *reinterpret_cast<const float4*>(my + 16 * 36 + (lane & 7) * 72 + ((it & 7) << 2));
      B1 = *reinterpret_cast<const float4*>(my + 16 * 36 + (lane & 7) * 72 + 36 + ((it & 7) << 2));
      C0 = *reinterpret_cast<const float4*>(my + 32 * 36 + (lane & 7) * 72 + ((it & 7) << 2));
      C1 = *reinterpret_cast<const float4*>(my + 32 * 36 + (lane & 7) * 72 + 36 + ((it & 7) << 2));
    } else {
      const float f = it * 1e-6f;
      dl4 = make_float4(0.01f + f, 0.02f, 0.03f + f, 0.04f); du4 = make_float4(1.f, 2.f + f, 3.f, 4.f);
      B0 = du4; B1 = dl4; C0 = make_float4(0.5f, 0.25f + f, 0.125f, 1.f); C1 = C0;
    }
    float y4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float dl = (&dl4.x)[e], du = (&du4.x)[e];
      float y = 0.f;
      const float a0 = __builtin_amdgcn_exp2f(dl * A2[0]);
      x[0] = fmaf(a0, x[0], du * (&B0.x)[e]);
      y = fmaf(x[0], (&C0.x)[e], y);
      const float a1 = __builtin_amdgcn_exp2f(dl * A2[1]);
      x[1] = fmaf(a1, x[1], du * (&B1.x)[e]);
      y = fmaf(x[1], (&C1.x)[e], y);
      if constexpr (MODE & 1) {
        y += dpp_f<0xB1>(y); y += dpp_f<0x4E>(y); y += dpp_f<0x141>(y);
      }
      y4[e] = y;
    }
    if constexpr (MODE & 4) {
      if ((lane & 7) == 0) *reinterpret_cast<float4*>(my + 8 * 36 + (lane >> 3) * 36 + ((it & 7) << 2)) = make_float4(y4[0], y4[1], y4[2], y4[3]);
    } else {
      acc += y4[0] + y4[1] + y4[2] + y4[3];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + x[0] + x[1] + my[lane];
}
template <int MODE>
void run(const char* name, float* d) {
  for (int wps = 1; wps <= 4; ++wps) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE><<<blocks, 256>>>(d, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    // per SIMD: wps waves x ITERS groups x 8 state-steps
    printf("%-22s waves/SIMD %d  %.3f ms -> %.2f ns per state-step-wave per SIMD\n", name, wps, ms, ms * 1e6 / (wps * ITERS * 8.0));
  }
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
  run<0>("mix", d); run<1>("mix+dpp", d); run<2>("mix+ldsread", d); run<3>("mix+dpp+ldsread", d); run<7>("mix+dpp+ldsrw", d);
  return 0;
}
