// Checks what `row_newbcast:n` does on gfx950 (expected: every lane reads lane n of its own 16-lane row) and what a
// v_fmac_f32_dpp with it costs next to a plain v_fmac_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dpp_newbcast.cpp -o /tmp/dpp_newbcast && /tmp/dpp_newbcast
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void sem(float* out) {
  const float v = (float)threadIdx.x;
  float r3, r12;
  asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(r3) : "v"(v));
  asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:12 row_mask:0xf bank_mask:0xf" : "=v"(r12) : "v"(v));
  out[threadIdx.x] = r3; out[64 + threadIdx.x] = r12;
}
template <bool DPP>
__global__ __launch_bounds__(256) void rate(float* out, float seed) {
  float b = seed + threadIdx.x, x0 = seed, x1 = seed * 2, x2 = seed * 3, x3 = seed * 4, a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int it = 0; it < 4096; ++it) {
    if constexpr (DPP) {
      asm volatile("v_fmac_f32_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %1, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %2, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %3, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(x0), "v"(x1), "v"(x2), "v"(x3));
    } else {
      asm volatile("v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %6\n\tv_fmac_f32 %2, %4, %7\n\tv_fmac_f32 %3, %4, %8"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(x0), "v"(x1), "v"(x2), "v"(x3));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
int main() {
  float* d; (void)hipMalloc(&d, 1024 * 256 * sizeof(float));
  sem<<<1, 64>>>(d);
  float h[128]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  bool ok = true;
  for (int i = 0; i < 64; ++i) ok = ok && h[i] == (float)((i & ~15) + 3) && h[64 + i] == (float)((i & ~15) + 12);
  printf("row_newbcast semantics (lane n of the own row): %s   lane 37 -> %g / %g\n", ok ? "OK" : "DIFFERENT", h[37], h[64 + 37]);
  for (int dpp = 0; dpp < 2; ++dpp)
    for (int wps = 1; wps <= 4; wps *= 2) {
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      if (dpp) rate<true><<<256 * wps, 256>>>(d, 0.5f); else rate<false><<<256 * wps, 256>>>(d, 0.5f);
      (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) { if (dpp) rate<true><<<256 * wps, 256>>>(d, 0.5f); else rate<false><<<256 * wps, 256>>>(d, 0.5f); }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      printf("%s waves/SIMD %d: %.3f ns per instruction per SIMD\n", dpp ? "v_fmac_f32_dpp row_newbcast" : "v_fmac_f32                 ", wps,
             ms * 1e6 / (wps * 4096.0 * 4));
    }
  return 0;
}
