// Diagnostic micro-benchmark for the lane = channel scan mapping: B/C as wave-uniform SGPR operands (scalar loads),
// delta', delta'*u from LDS (b128 per 4 steps), y partials to LDS with ds_add_f32; NS states per lane.
// Also: does a wave with only the low 32 lanes enabled issue VALU faster (EXEC-half skipping)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/sgpr_rate.cpp -o /tmp/sgpr_rate && /tmp/sgpr_rate
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 2048;   // 4-step groups per wave
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4f* cptr4;

// MODE bits: 1 = ds_add_f32 of y per step (else register accumulate), 2 = LDS b128 reads of dl/du, 4 = SGPR B/C via scalar loads
//            8 = half EXEC (low 32 lanes only)
template <int NS, int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ bc, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[4][64 * 36 + 32 * 65];
  float* my = lds[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  for (int i = lane; i < 64 * 36 + 32 * 65; i += 64) my[i] = seed + i * 1e-6f;
  float x[NS], A2[NS], acc = 0.f;
#pragma unroll
  for (int j = 0; j < NS; ++j) { x[j] = seed * (j + 1); A2[j] = -1.f - j - lane * 1e-3f; }
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  cptr4 pbc = (cptr4)(bc + (size_t)(blockIdx.x & 63) * 4096 + wv * 512);
  float* yacc = my + 64 * 36;
  if constexpr (MODE & 8) { if (lane >= 32) { out[blockIdx.x * blockDim.x + threadIdx.x] = 0.f; return; } }
  for (int it = 0; it < ITERS; ++it) {
    v4f dl4, du4, Bv[NS], Cv[NS];
    if constexpr (MODE & 2) {
      const float* base = my + lane * 36 + ((it & 7) << 2);
      dl4 = *reinterpret_cast<const v4f*>(base);
      du4 = *reinterpret_cast<const v4f*>(base + 32 * 36 - (lane >= 32 ? 32 * 36 : 0));
    } else {
      const float f = it * 1e-6f;
      dl4 = (v4f){0.01f + f, 0.02f, 0.03f + f, 0.04f}; du4 = (v4f){1.f, 2.f + f, 3.f, 4.f};
    }
    if constexpr (MODE & 4) {
#pragma unroll
      for (int j = 0; j < NS; ++j) { Bv[j] = pbc[(it & 15) * 2 * NS + j]; Cv[j] = pbc[(it & 15) * 2 * NS + NS + j]; }
    } else {
#pragma unroll
      for (int j = 0; j < NS; ++j) { Bv[j] = du4 + (float)j; Cv[j] = dl4 * (float)(j + 1); }
    }
    float y4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float y = 0.f;
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const float a = __builtin_amdgcn_exp2f(dl4[e] * A2[j]);
        x[j] = fmaf(a, x[j], du4[e] * Bv[j][e]);
        y = fmaf(x[j], Cv[j][e], y);
      }
      y4[e] = y;
    }
    if constexpr (MODE & 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(yacc + (((it & 7) << 2) + e) * 65 + lane, y4[e]);
    } else {
      acc += y4[0] + y4[1] + y4[2] + y4[3];
    }
  }
  float r = acc + my[lane] + yacc[lane];
#pragma unroll
  for (int j = 0; j < NS; ++j) r += x[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NS, int MODE>
void run(const char* name, float* d, const float* bc) {
  for (int wps = 1; wps <= 4; ++wps) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NS, MODE><<<blocks, 256>>>(d, bc, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<NS, MODE><<<blocks, 256>>>(d, bc, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-28s NS=%d waves/SIMD %d  %.3f ms -> %.2f ns per wave-state-step per SIMD (floor ~5.3)\n", name, NS, wps, ms,
           ms * 1e6 / (wps * ITERS * 4.0 * NS));
  }
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
  float* bc; (void)hipMalloc(&bc, 64 * 4096 * sizeof(float) + 65536); (void)hipMemset(bc, 0, 64 * 4096 * sizeof(float) + 65536);
  run<4, 0>("mix regs only", d, bc); run<4, 4>("mix + sgpr B/C", d, bc); run<4, 6>("mix + sgpr + ldsread", d, bc);
  run<4, 7>("mix + sgpr + ldsread + dsadd", d, bc); run<2, 7>("mix + sgpr + ldsread + dsadd", d, bc);
  run<8, 7>("mix + sgpr + ldsread + dsadd", d, bc);
  run<4, 8>("regs only, half EXEC", d, bc); run<4, 15>("full mix, half EXEC", d, bc);
  return 0;
}
