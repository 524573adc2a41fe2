// Sum over the leading dimension of a dense fp32 tensor: dst[i] = sum_{j < nlead} src[j * lead_stride + i], i < ninner.
// What it replaces: the `.sum(0)` behind every batched weight-gradient GEMM of the batch-major stages (MedMamba.py:259, 262, 292,
// 302 differentiated: dW = sum_b dY_b X_b^T), behind the per-workgroup partial rows of the LayerNorm / split kernels and behind the
// per-image partial products of the deterministic conv weight gradient — ~50 ATen reductions of 1-3 MB per training step, each a
// 10 us launch of a generic reduce kernel.  Here: 64 inner elements (as float4) x 4 lead parts per 256-thread workgroup, every
// part's loads in flight 8 at a time, the 4 partial sums joined through LDS in a fixed order (deterministic, no atomics).
#include "mm_common.h"
#include "medmamba_hip.h"

namespace {
using namespace mm;

template <bool VEC>
__global__ __launch_bounds__(256) void sum_lead_kernel(const float* __restrict__ src, float* __restrict__ dst, int nlead, int64_t ninner,
                                                       int64_t lead_stride) {
  constexpr int W = VEC ? 4 : 1;                       // floats per thread
  __shared__ float part[4][64][W];
  const int col = threadIdx.x & 63, jp = threadIdx.x >> 6;          // 64 inner slots x 4 lead parts
  const int64_t i0 = ((int64_t)blockIdx.x * 64 + col) * W;
  const bool ok = i0 < ninner;
  const int per = (nlead + 3) >> 2, j0 = jp * per, j1 = min(nlead, j0 + per);
  float acc[W];
#pragma unroll
  for (int w = 0; w < W; ++w) acc[w] = 0.f;
  const float* p = src + i0;
  for (int j = j0; j < j1; j += 8) {
    float v[8][W];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool in = ok && j + u < j1;
      if constexpr (VEC) {
        const float4 t = in ? *reinterpret_cast<const float4*>(p + (int64_t)(j + u) * lead_stride) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
      } else {
        v[u][0] = in ? p[(int64_t)(j + u) * lead_stride] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int w = 0; w < W; ++w) acc[w] += v[u][w];
  }
#pragma unroll
  for (int w = 0; w < W; ++w) part[jp][col][w] = acc[w];
  __syncthreads();
  if (jp == 0 && ok) {
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const float s = (part[0][col][w] + part[1][col][w]) + (part[2][col][w] + part[3][col][w]);
      if (i0 + w < ninner) dst[i0 + w] = s;
    }
  }
}
}  // namespace

extern "C" int mm_sum_lead(const float* src, float* dst, int nlead, int64_t ninner, int64_t lead_stride, void* stream) {
  if (!src || !dst) return MM_ERR_NULL;
  if (nlead <= 0 || ninner <= 0 || lead_stride < ninner) return MM_ERR_SHAPE;
  const bool vec = ninner % 4 == 0 && lead_stride % 4 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  const int64_t slots = vec ? ninner / 4 : ninner;
  const int64_t nb = (slots + 63) / 64;
  if (nb > 0x7fffffffll) return MM_ERR_SHAPE;
  if (vec) hipLaunchKernelGGL(sum_lead_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, dst, nlead, ninner, lead_stride);
  else hipLaunchKernelGGL(sum_lead_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, dst, nlead, ninner, lead_stride);
  return (int)hipGetLastError();
}
