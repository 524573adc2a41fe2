// Sum over the leading dimension of a dense fp32 tensor: dst[i] = sum_{j < nlead} src[j * lead_stride + i], i < ninner.
// What it replaces: the `.sum(0)` behind every batched weight-gradient GEMM of the batch-major stages (MedMamba.py:259, 262, 292,
// 302 differentiated: dW = sum_b dY_b X_b^T), behind the per-workgroup partial rows of the LayerNorm / split kernels and behind the
// per-image partial products of the deterministic conv weight gradient — ~50 ATen reductions of 1-3 MB per training step, each a
// 10 us launch of a generic reduce kernel.  Here: SX inner slots (float4 each) x 256/SX interleaved lead parts per 256-thread
// workgroup, 8 loads in flight per thread, the partial sums joined through LDS in a fixed order (deterministic, no atomics).  Meant
// for lead dimensions of a batch or of the partial rows of a kernel (<= 4096 rows: the callers leave taller reductions to ATen).
#include "mm_common.h"
#include "medmamba_hip.h"

namespace {
using namespace mm;

template <bool VEC>
__global__ __launch_bounds__(256) void sum_lead_kernel(const float* __restrict__ src, float* __restrict__ dst, int nlead, int64_t ninner,
                                                       int64_t lead_stride, int sx_log2, int64_t chunk, int64_t dst_chunk_stride) {
  constexpr int W = VEC ? 4 : 1;                       // floats per thread
  __shared__ float part[256][W];
  // 256 threads = SX inner slots x PY lead parts (SX = 64 for wide tensors, fewer slots -> more parts for narrow ones); part p
  // takes rows p, p + PY, p + 2 PY, ... — 8 loads in flight per thread
  const int SX = 1 << sx_log2, PY = 256 >> sx_log2;
  const int col = threadIdx.x & (SX - 1), jp = threadIdx.x >> sx_log2;
  const int64_t i0 = ((int64_t)blockIdx.x * SX + col) * W;
  const bool ok = i0 < ninner;
  float acc[W];
#pragma unroll
  for (int w = 0; w < W; ++w) acc[w] = 0.f;
  const float* p = src + (ok ? i0 : 0);
  for (int j = jp; j < nlead; j += 8 * PY) {
    float v[8][W];
    // unconditional loads from a clamped (always valid) row, the value dropped afterwards: a load behind `in ? ... : 0` is a branch
    // around the load and a wait before the next one — one load in flight per thread instead of eight
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int jj = j + u * PY;
      const bool in = ok && jj < nlead;
      const int64_t row = (int64_t)(jj < nlead ? jj : nlead - 1) * lead_stride;
      if constexpr (VEC) {
        const float4 t = *reinterpret_cast<const float4*>(p + row);
        v[u][0] = in ? t.x : 0.f; v[u][1] = in ? t.y : 0.f; v[u][2] = in ? t.z : 0.f; v[u][3] = in ? t.w : 0.f;
      } else {
        const float t = p[row];
        v[u][0] = in ? t : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int w = 0; w < W; ++w) acc[w] += v[u][w];
  }
#pragma unroll
  for (int w = 0; w < W; ++w) part[threadIdx.x][w] = acc[w];
  __syncthreads();
  if (jp == 0 && ok) {                                 // the parts in a fixed order
#pragma unroll
    for (int w = 0; w < W; ++w) {
      float s = part[col][w];
      for (int q = 1; q < PY; ++q) s += part[q * SX + col][w];
      // chunked destination (mm_sum_lead_chunks): element i of the dense source row lies at (i / chunk) * stride + i % chunk
      if (i0 + w < ninner) dst[chunk > 0 ? ((i0 + w) / chunk) * dst_chunk_stride + (i0 + w) % chunk : i0 + w] = s;
    }
  }
}
}  // namespace

namespace {
int sum_lead_launch(const float* src, float* dst, int nlead, int64_t ninner, int64_t lead_stride, int64_t chunk, int64_t dst_chunk_stride,
                    void* stream) {
  if (!src || !dst) return MM_ERR_NULL;
  if (nlead <= 0 || ninner <= 0 || lead_stride < ninner) return MM_ERR_SHAPE;
  if (chunk > 0 && (ninner % chunk != 0 || dst_chunk_stride < chunk)) return MM_ERR_SHAPE;
  const bool vec = ninner % 4 == 0 && lead_stride % 4 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 &&
                   (chunk <= 0 || (chunk % 4 == 0 && dst_chunk_stride % 4 == 0));
  const int64_t slots = vec ? ninner / 4 : ninner;
  int sx_log2 = 6;                                      // 64 slots per workgroup; narrow tensors: fewer slots, more lead parts;
  while (sx_log2 > 3 && (int64_t)(1 << (sx_log2 - 1)) >= slots) --sx_log2;
  // tall tensors (partial rows of a thousand workgroups): at most 32 rows per part = four rounds of eight loads, whatever the width
  while (sx_log2 > 3 && nlead > (256 >> sx_log2) * 32) --sx_log2;
  const int64_t nb = (slots + (1 << sx_log2) - 1) >> sx_log2;
  if (nb > 0x7fffffffll) return MM_ERR_SHAPE;
  if (vec) hipLaunchKernelGGL(sum_lead_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, dst, nlead, ninner, lead_stride, sx_log2, chunk, dst_chunk_stride);
  else hipLaunchKernelGGL(sum_lead_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, dst, nlead, ninner, lead_stride, sx_log2, chunk, dst_chunk_stride);
  return (int)hipGetLastError();
}
}  // namespace

extern "C" int mm_sum_lead(const float* src, float* dst, int nlead, int64_t ninner, int64_t lead_stride, void* stream) {
  return sum_lead_launch(src, dst, nlead, ninner, lead_stride, 0, 0, stream);
}

extern "C" int mm_sum_lead_chunks(const float* src, float* dst, int nlead, int64_t nchunks, int64_t chunk, int64_t dst_chunk_stride,
                                  void* stream) {
  if (nchunks <= 0 || chunk <= 0) return MM_ERR_SHAPE;
  return sum_lead_launch(src, dst, nlead, nchunks * chunk, nchunks * chunk, chunk, dst_chunk_stride, stream);
}
