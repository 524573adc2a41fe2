// Streaming read/write rates with and without the non-temporal hint on buffer loads / stores (gfx950: aux bit 1 = nt, bit 0 = sc0,
// bit 4 = sc1).  out[i] = a[i] + b[i] over float4; sizes like the plane sets of the 14x14 / 56x56 stages.
// build: hipcc --offload-arch=gfx950 -O3 -o stream_nt stream_nt.hip ; run: ./stream_nt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int LA, int SA, int NIN>
__global__ __launch_bounds__(256) void k(const float* a, const float* b, float* o, long n4, int per) {
  // every workgroup owns a contiguous range of `per` float4 per thread-stride chunk
  const long base = (long)blockIdx.x * per * 256;
  for (int i = 0; i < per; i += 4) {
    v4f va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long idx = base + (long)(i + u) * 256 + threadIdx.x;
      const bool ok = idx < n4;
      const rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(a + base * 4), 0, 0x7ffffff0, 0x00020000);
      const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(b + base * 4), 0, 0x7ffffff0, 0x00020000);
      const int off = ok ? (int)((idx - base) * 16) : 0x7fffffff;
      va[u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(ra, off, 0, LA));
      if (NIN > 1) vb[u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rb, off, 0, LA)); else vb[u] = va[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long idx = base + (long)(i + u) * 256 + threadIdx.x;
      const bool ok = idx < n4;
      const rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(o + base * 4), 0, 0x7ffffff0, 0x00020000);
      const int off = ok ? (int)((idx - base) * 16) : 0x7fffffff;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, va[u] + vb[u]), ro, off, 0, SA);
    }
  }
}

template <int LA, int SA, int NIN>
float run(const float* a, const float* b, float* o, long n4) {
  const int per = 16;
  const int grid = (int)((n4 + 256L * per - 1) / (256L * per));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<LA, SA, NIN>), dim3(grid), dim3(256), 0, 0, a, b, o, n4, per);
  hipEventRecord(e0);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k<LA, SA, NIN>), dim3(grid), dim3(256), 0, 0, a, b, o, n4, per);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / it * 1e3f;
}

int main() {
  for (long mb : {19L, 77L, 308L}) {
    const long n = mb * 1000 * 1000 / 4, n4 = n / 4;
    float *a, *b, *o;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&o, n * 4);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
    const double gb2 = 3.0 * n * 4 / 1e3, gb1 = 2.0 * n * 4 / 1e3;   // bytes / us -> MB/us = TB/s * 1e3... printed as GB/s
    float t;
#define R(LA, SA, NIN, name) t = run<LA, SA, NIN>(a, b, o, n4); printf("%4ld MB planes  %-28s %8.1f us  %7.0f GB/s\n", mb, name, t, (NIN > 1 ? gb2 : gb1) / t);
    R(0, 0, 2, "2 in 1 out plain")
    R(0, 2, 2, "2 in 1 out nt store")
    R(2, 0, 2, "2 in 1 out nt load")
    R(2, 2, 2, "2 in 1 out nt both")
    R(0, 17, 2, "2 in 1 out sc0sc1 store")
    R(0, 19, 2, "2 in 1 out sc0sc1nt store")
    R(0, 0, 1, "1 in 1 out plain")
    R(2, 2, 1, "1 in 1 out nt both")
    hipFree(a); hipFree(b); hipFree(o);
  }
  return 0;
}
