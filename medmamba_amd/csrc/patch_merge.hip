// PatchMerging2D (MedMamba.py:79-119) front half in one pass over HBM: the 2x2 strided gather with the reference's
// channel order (x0 = even row / even col, x1 = odd row / even col, x2 = even row / odd col, x3 = odd / odd; odd H or W are
// cropped to the even part, MedMamba.py:97-111) + LayerNorm over the 4C gathered channels.  The Linear(4C -> 2C) that
// follows stays a library GEMM on the rows this kernel writes.
//
// One wavefront per output position; a row of 4C floats = C float4, lane l holds float4 l, l+64, ... (<= 8 of them:
// 4C <= 2048) in registers between the statistics and the output pass, so every operand crosses HBM once (ATen's
// LayerNorm on a materialised gather reads / writes the 4C-wide rows three times and ran below 0.4 TB/s on them).
// Exact two-pass statistics (mean, then centred sum of squares) like torch.nn.LayerNorm.
// Backward: d(x) is written straight into the gathered positions of d(input) (the cropped last row / column of an odd
// image is zero-filled by the caller), dgamma / dbeta leave as one partial row per wavefront (the caller sums them).
#include "mm_common.h"
#include "medmamba_hip.h"

namespace {
using namespace mm;

__device__ __forceinline__ float wave_sum(float v) {
  v = group_sum<16>(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

struct PmGeom { int H, W, C, h2, w2, C4; };   // C4 = C / 4 float4 per source pixel

__device__ __forceinline__ float ldb1(rsrc_t r, int off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ float4 ldb4(rsrc_t r, int off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void stb4(rsrc_t r, int off, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, off, 0, 0);
}
// Round 4: the rows of these kernels go through buffer descriptors that start at the row (wave-uniform 64-bit base, 32-bit lane
// offsets, kOOB for lanes beyond the row): no branch around a load, so the NV (x 2) loads of a lane are in flight together; the
// position of a lane's float4 inside the 2x2 gather does not depend on the row and is computed once; the row index is wave-uniform
// (scalar divisions); gamma / beta stay in registers.  (Before: one conditional load at a time, three 64-bit vector divisions per
// float4, 4 wavefronts per CU — 2.2 TB/s on the 56x56 -> 28x28 merge.)
// byte offset of float4 j of ANY output row relative to the row's first source pixel (2*h2i, 2*w2i)
__device__ __forceinline__ int pm_lane_off(const PmGeom& g, int j) {
  const int s = j / g.C4, o = j - s * g.C4;
  return (((s & 1) * g.W + (s >> 1)) * g.C4 + o) * 16;
}
// float4 index of the first source pixel of output row `row` (wave-uniform)
__device__ __forceinline__ int64_t pm_row_base(const PmGeom& g, int64_t row) {
  const int w2i = (int)(row % g.w2);
  const int64_t t = row / g.w2;
  const int h2i = (int)(t % g.h2);
  const int64_t b = t / g.h2;
  return ((b * g.H + 2 * h2i) * g.W + 2 * w2i) * (int64_t)g.C4;
}

template <int NV>
__global__ __launch_bounds__(256) void patch_merge_ln_fwd_kernel(const float4* __restrict__ x, const float4* __restrict__ gamma,
                                                                 const float4* __restrict__ beta, float eps, float4* __restrict__ out,
                                                                 float* __restrict__ mu_out, float* __restrict__ rstd_out,
                                                                 int64_t nrows, PmGeom g) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  const float inv_n = 1.f / (4 * g.C);
  const rsrc_t rgam = make_rsrc(gamma, (int64_t)g.C * 16), rbet = make_rsrc(beta, (int64_t)g.C * 16);
  const int64_t span = ((int64_t)(g.W + 1) * g.C4 + g.C4) * 16;      // two pixel rows of the 2x2 gather, from its first pixel
  float4 gm[NV], bt[NV];
  int xoff[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int j = lane + 64 * k;
    gm[k] = ldb4(rgam, j * 16); bt[k] = ldb4(rbet, j * 16);          // beyond the row: 0
    xoff[k] = j < g.C ? pm_lane_off(g, j) : kOOB;
  }
  for (int64_t row = wave_global; row < nrows; row += nwaves) {
    const rsrc_t rx = make_rsrc(x + pm_row_base(g, row), span), ro = make_rsrc(out + row * g.C, (int64_t)g.C * 16);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = ldb4(rx, xoff[k]);
#pragma unroll
    for (int k = 0; k < NV; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    const float mean = wave_sum(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (lane + 64 * k < g.C) {
        const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
    }
    const float rstd = __builtin_amdgcn_rsqf(wave_sum(q) * inv_n + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k)
      stb4(ro, (lane + 64 * k) * 16, make_float4((v[k].x - mean) * rstd * gm[k].x + bt[k].x, (v[k].y - mean) * rstd * gm[k].y + bt[k].y,
                                               (v[k].z - mean) * rstd * gm[k].z + bt[k].z, (v[k].w - mean) * rstd * gm[k].w + bt[k].w));
    if (lane == 0) { mu_out[row] = mean; rstd_out[row] = rstd; }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  ws[wave][0: dgamma | 4C: dbeta]
template <int NV>
__global__ __launch_bounds__(256) void patch_merge_ln_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ x,
                                                                 const float4* __restrict__ gamma, const float* __restrict__ mu_in,
                                                                 const float* __restrict__ rstd_in, float4* __restrict__ dinp,
                                                                 float4* __restrict__ ws, int64_t nrows, PmGeom g) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  const float inv_n = 1.f / (4 * g.C);
  const rsrc_t rgam = make_rsrc(gamma, (int64_t)g.C * 16);
  const int64_t span = ((int64_t)(g.W + 1) * g.C4 + g.C4) * 16;      // two pixel rows of the 2x2 gather, from its first pixel
  float4 ag[NV], ab[NV], gm[NV];
  int xoff[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int j = lane + 64 * k;
    ag[k] = make_float4(0.f, 0.f, 0.f, 0.f); ab[k] = ag[k];
    gm[k] = ldb4(rgam, j * 16);                                      // beyond the row: 0
    xoff[k] = j < g.C ? pm_lane_off(g, j) : kOOB;
  }
  for (int64_t row = wave_global; row < nrows; row += nwaves) {
    const int64_t rb = pm_row_base(g, row);
    const rsrc_t rx = make_rsrc(x + rb, span), rdx = make_rsrc(dinp + rb, span), rdy = make_rsrc(dy + row * g.C, (int64_t)g.C * 16);
    const float mean = mu_in[row], rstd = rstd_in[row];
    float4 xh[NV], gg[NV], dv[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) { xh[k] = ldb4(rx, xoff[k]); dv[k] = ldb4(rdy, (lane + 64 * k) * 16); }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const bool in = lane + 64 * k < g.C;
      const float4 xv = xh[k], d = dv[k];                            // d = 0 beyond the row: every product below vanishes
      xh[k] = in ? make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
      gg[k] = make_float4(d.x * gm[k].x, d.y * gm[k].y, d.z * gm[k].z, d.w * gm[k].w);
      s1 += (gg[k].x + gg[k].y) + (gg[k].z + gg[k].w);
      s2 += (gg[k].x * xh[k].x + gg[k].y * xh[k].y) + (gg[k].z * xh[k].z + gg[k].w * xh[k].w);
      ag[k].x = fmaf(d.x, xh[k].x, ag[k].x); ag[k].y = fmaf(d.y, xh[k].y, ag[k].y);
      ag[k].z = fmaf(d.z, xh[k].z, ag[k].z); ag[k].w = fmaf(d.w, xh[k].w, ag[k].w);
      ab[k].x += d.x; ab[k].y += d.y; ab[k].z += d.z; ab[k].w += d.w;
    }
    const float m1 = wave_sum(s1) * inv_n, m2 = wave_sum(s2) * inv_n;
#pragma unroll
    for (int k = 0; k < NV; ++k)
      stb4(rdx, xoff[k], make_float4(rstd * (gg[k].x - m1 - xh[k].x * m2), rstd * (gg[k].y - m1 - xh[k].y * m2),
                                     rstd * (gg[k].z - m1 - xh[k].z * m2), rstd * (gg[k].w - m1 - xh[k].w * m2)));
  }
  float4* wrow = ws + wave_global * 2 * g.C;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int j = lane + 64 * k;
    if (j < g.C) { wrow[j] = ag[k]; wrow[g.C + j] = ab[k]; }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// PatchEmbed2D's back half (MedMamba.py:70-76): the strided conv's NCHW output -> NHWC rows + LayerNorm(C), one pass each way.
// ATen runs it as a permute copy + LayerNorm (+ two more copies and a two-kernel LayerNorm backward): 0.27 ms forward, 0.49 ms
// backward per step at 64 x 96 x 56 x 56.  Here a workgroup moves a tile of kPeT positions x C channels through LDS: planes are
// read / written along positions (128-B runs), rows along channels; a wavefront normalises one position at a time, lane l
// holding channels l, l + 64, ...  (C <= 512).  Exact two-pass statistics like torch.nn.LayerNorm.
constexpr int kPeT = 32;            // positions per tile
constexpr int kPeS = kPeT + 1;      // LDS row stride (floats): column reads across channels hit distinct banks

template <int NV>
__global__ __launch_bounds__(256) void nchw_ln_rows_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, float* __restrict__ out,
                                                               float* __restrict__ mu_out, float* __restrict__ rstd_out, int C,
                                                               int HW, int tiles_per_img, int64_t ntiles) {
  extern __shared__ float pe_tile[];            // [C][kPeS]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float inv_n = 1.f / C;
  float gm[NV], bt[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = lane + 64 * k;
    gm[k] = c < C ? gamma[c] : 0.f;
    bt[k] = c < C ? beta[c] : 0.f;
  }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b = tile / tiles_per_img;
    const int p0 = (int)(tile - b * tiles_per_img) * kPeT;
    const int np = min(kPeT, HW - p0);
    const rsrc_t rt = make_rsrc(x + b * (int64_t)C * HW + p0, ((int64_t)(C - 1) * HW + np) * 4);     // this tile's C rows of np positions
    __syncthreads();                            // the previous tile's readers are done
    for (int idx0 = tid; idx0 < C * kPeT; idx0 += 4 * 256) {       // four loads of a thread in flight together, no branch around them
      float t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = idx0 + u * 256, c = idx / kPeT, pp = idx % kPeT;
        t[u] = ldb1(rt, (idx < C * kPeT && pp < np) ? (c * HW + pp) * 4 : kOOB);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = idx0 + u * 256;
        if (idx < C * kPeT) pe_tile[(idx / kPeT) * kPeS + idx % kPeT] = t[u];
      }
    }
    __syncthreads();
    for (int pp = wave; pp < np; pp += 4) {
      float v[NV];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        v[k] = c < C ? pe_tile[c * kPeS + pp] : 0.f;
        s += v[k];
      }
      const float mean = wave_sum(s) * inv_n;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const float d = lane + 64 * k < C ? v[k] - mean : 0.f;
        q = fmaf(d, d, q);
      }
      const float rstd = __builtin_amdgcn_rsqf(wave_sum(q) * inv_n + eps);
      const int64_t row = b * HW + p0 + pp;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < C) out[row * C + c] = (v[k] - mean) * rstd * gm[k] + bt[k];
      }
      if (lane == 0) { mu_out[row] = mean; rstd_out[row] = rstd; }
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma, written back as NCHW planes;
// ws[block][0 : C] = d(gamma) partial, ws[block][C : 2C] = d(beta) partial
template <int NV>
__global__ __launch_bounds__(256) void nchw_ln_rows_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ gamma, const float* __restrict__ mu_in,
                                                               const float* __restrict__ rstd_in, float* __restrict__ dx,
                                                               float* __restrict__ ws, int C, int HW, int tiles_per_img,
                                                               int64_t ntiles) {
  extern __shared__ float pe_tile[];            // [C][kPeS], then [4][2C] for the final reduction
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float inv_n = 1.f / C;
  float gm[NV], ag[NV], ab[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = lane + 64 * k;
    gm[k] = c < C ? gamma[c] : 0.f;
    ag[k] = 0.f; ab[k] = 0.f;
  }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b = tile / tiles_per_img;
    const int p0 = (int)(tile - b * tiles_per_img) * kPeT;
    const int np = min(kPeT, HW - p0);
    const int64_t base = b * (int64_t)C * HW + p0;
    const rsrc_t rt = make_rsrc(x + base, ((int64_t)(C - 1) * HW + np) * 4);
    __syncthreads();
    for (int idx0 = tid; idx0 < C * kPeT; idx0 += 4 * 256) {
      float t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = idx0 + u * 256, c = idx / kPeT, pp = idx % kPeT;
        t[u] = ldb1(rt, (idx < C * kPeT && pp < np) ? (c * HW + pp) * 4 : kOOB);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = idx0 + u * 256;
        if (idx < C * kPeT) pe_tile[(idx / kPeT) * kPeS + idx % kPeT] = t[u];
      }
    }
    __syncthreads();
    for (int pp = wave; pp < np; pp += 4) {
      const int64_t row = b * HW + p0 + pp;
      const float mean = mu_in[row], rstd = rstd_in[row];
      const rsrc_t rdy = make_rsrc(dy + row * C, (int64_t)C * 4);
      float xh[NV], gg[NV], dv[NV];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) dv[k] = ldb1(rdy, (lane + 64 * k) * 4);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < C) {
          const float d = dv[k];
          xh[k] = (pe_tile[c * kPeS + pp] - mean) * rstd;
          gg[k] = d * gm[k];
          s1 += gg[k];
          s2 = fmaf(gg[k], xh[k], s2);
          ag[k] = fmaf(d, xh[k], ag[k]);
          ab[k] += d;
        } else {
          xh[k] = 0.f; gg[k] = 0.f;
        }
      }
      const float m1 = wave_sum(s1) * inv_n, m2 = wave_sum(s2) * inv_n;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < C) pe_tile[c * kPeS + pp] = rstd * (gg[k] - m1 - xh[k] * m2);   // this wave owns column pp
      }
    }
    __syncthreads();
    for (int idx = tid; idx < C * kPeT; idx += 256) {
      const int c = idx / kPeT, pp = idx % kPeT;
      if (pp < np) dx[base + (int64_t)c * HW + pp] = pe_tile[c * kPeS + pp];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int c = lane + 64 * k;
    if (c < C) { pe_tile[wave * 2 * C + c] = ag[k]; pe_tile[wave * 2 * C + C + c] = ab[k]; }
  }
  __syncthreads();
  for (int j = tid; j < 2 * C; j += 256)
    ws[(int64_t)blockIdx.x * 2 * C + j] = (pe_tile[j] + pe_tile[2 * C + j]) + (pe_tile[4 * C + j] + pe_tile[6 * C + j]);
}

inline int pe_grid(int64_t ntiles) { return (int)(ntiles < 1 ? 1 : (ntiles > 2048 ? 2048 : ntiles)); }

inline int pm_grid(int64_t nrows) {   // >= 4 rows per wavefront, at most 1024 workgroups: 16 wavefronts per CU (4 per CU left the loads
  int64_t b = (nrows + 15) / 16;      // of a wavefront's row as the only ones in flight), 4096 partial dgamma / dbeta rows at most
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

extern "C" {

int mm_patch_merge_ln_supported(int C) { return (C > 0 && C % 4 == 0 && C <= 512) ? 1 : 0; }
int mm_patch_merge_ln_rows(int batch, int H, int W) { return 4 * pm_grid((int64_t)batch * (H / 2) * (W / 2)); }

#define MM_PM_DISPATCH(KERNEL, ...)                                                                                   \
  do {                                                                                                                \
    const int nv = (g.C + 63) / 64;                                                                                   \
    const dim3 grid(pm_grid(nrows)), block(256);                                                                      \
    if (nv <= 1) hipLaunchKernelGGL(KERNEL<1>, grid, block, 0, s, __VA_ARGS__);                                       \
    else if (nv <= 2) hipLaunchKernelGGL(KERNEL<2>, grid, block, 0, s, __VA_ARGS__);                                  \
    else if (nv <= 3) hipLaunchKernelGGL(KERNEL<3>, grid, block, 0, s, __VA_ARGS__);   /* 4C = 768: no idle slot */   \
    else if (nv <= 4) hipLaunchKernelGGL(KERNEL<4>, grid, block, 0, s, __VA_ARGS__);                                  \
    else if (nv <= 6) hipLaunchKernelGGL(KERNEL<6>, grid, block, 0, s, __VA_ARGS__);   /* 4C = 1536 */                \
    else hipLaunchKernelGGL(KERNEL<8>, grid, block, 0, s, __VA_ARGS__);                                               \
  } while (0)

int mm_patch_merge_ln_fwd(const float* x, const float* gamma, const float* beta, float eps, float* out, float* mu, float* rstd,
                          int batch, int H, int W, int C, void* stream) {
  if (!x || !gamma || !beta || !out || !mu || !rstd) return MM_ERR_NULL;
  if (batch <= 0 || H < 2 || W < 2 || C <= 0) return MM_ERR_SHAPE;
  if (!mm_patch_merge_ln_supported(C)) return MM_ERR_UNSUPPORTED;
  if (!al16(x) || !al16(gamma) || !al16(beta) || !al16(out)) return MM_ERR_ALIGN;
  const PmGeom g{H, W, C, H / 2, W / 2, C / 4};
  const int64_t nrows = (int64_t)batch * g.h2 * g.w2;
  hipStream_t s = (hipStream_t)stream;
  MM_PM_DISPATCH(patch_merge_ln_fwd_kernel, (const float4*)x, (const float4*)gamma, (const float4*)beta, eps, (float4*)out, mu, rstd,
                 nrows, g);
  return (int)hipGetLastError();
}

int mm_patch_merge_ln_bwd(const float* dy, const float* x, const float* gamma, const float* mu, const float* rstd, float* dinp,
                          float* ws, int batch, int H, int W, int C, void* stream) {
  if (!dy || !x || !gamma || !mu || !rstd || !dinp || !ws) return MM_ERR_NULL;
  if (batch <= 0 || H < 2 || W < 2 || C <= 0) return MM_ERR_SHAPE;
  if (!mm_patch_merge_ln_supported(C)) return MM_ERR_UNSUPPORTED;
  if (!al16(dy) || !al16(x) || !al16(gamma) || !al16(dinp) || !al16(ws)) return MM_ERR_ALIGN;
  const PmGeom g{H, W, C, H / 2, W / 2, C / 4};
  const int64_t nrows = (int64_t)batch * g.h2 * g.w2;
  hipStream_t s = (hipStream_t)stream;
  MM_PM_DISPATCH(patch_merge_ln_bwd_kernel, (const float4*)dy, (const float4*)x, (const float4*)gamma, mu, rstd, (float4*)dinp,
                 (float4*)ws, nrows, g);
  return (int)hipGetLastError();
}

int mm_nchw_ln_rows_supported(int C) { return (C > 0 && C <= 512) ? 1 : 0; }
int mm_nchw_ln_rows_ws_rows(int batch, int HW) { return pe_grid((int64_t)batch * ((HW + kPeT - 1) / kPeT)); }

#define MM_PE_DISPATCH(KERNEL, ...)                                                                                   \
  do {                                                                                                                \
    const int nv = (C + 63) / 64;                                                                                     \
    const size_t lds = sizeof(float) * (size_t)C * kPeS;                                                              \
    const dim3 grid(pe_grid(ntiles)), block(256);                                                                     \
    if (nv <= 2) { if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)KERNEL<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                   hipLaunchKernelGGL(KERNEL<2>, grid, block, lds, s, __VA_ARGS__); }                                 \
    else if (nv <= 4) { if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)KERNEL<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                        hipLaunchKernelGGL(KERNEL<4>, grid, block, lds, s, __VA_ARGS__); }                            \
    else { if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)KERNEL<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
           hipLaunchKernelGGL(KERNEL<8>, grid, block, lds, s, __VA_ARGS__); }                                         \
  } while (0)

int mm_nchw_ln_rows_fwd(const float* x, const float* gamma, const float* beta, float eps, float* out, float* mu, float* rstd,
                        int batch, int C, int HW, void* stream) {
  if (!x || !gamma || !beta || !out || !mu || !rstd) return MM_ERR_NULL;
  if (batch <= 0 || C <= 0 || HW <= 0) return MM_ERR_SHAPE;
  if (!mm_nchw_ln_rows_supported(C)) return MM_ERR_UNSUPPORTED;
  if ((int64_t)C * HW * 4 >= 0x7fffffffll) return MM_ERR_UNSUPPORTED;      // 32-bit byte offsets inside one image (C planes of HW)
  const int tiles_per_img = (HW + kPeT - 1) / kPeT;
  const int64_t ntiles = (int64_t)batch * tiles_per_img;
  hipStream_t s = (hipStream_t)stream;
  MM_PE_DISPATCH(nchw_ln_rows_fwd_kernel, x, gamma, beta, eps, out, mu, rstd, C, HW, tiles_per_img, ntiles);
  return (int)hipGetLastError();
}

int mm_nchw_ln_rows_bwd(const float* dy, const float* x, const float* gamma, const float* mu, const float* rstd, float* dx,
                        float* ws, int batch, int C, int HW, void* stream) {
  if (!dy || !x || !gamma || !mu || !rstd || !dx || !ws) return MM_ERR_NULL;
  if (batch <= 0 || C <= 0 || HW <= 0) return MM_ERR_SHAPE;
  if (!mm_nchw_ln_rows_supported(C)) return MM_ERR_UNSUPPORTED;
  if ((int64_t)C * HW * 4 >= 0x7fffffffll) return MM_ERR_UNSUPPORTED;
  const int tiles_per_img = (HW + kPeT - 1) / kPeT;
  const int64_t ntiles = (int64_t)batch * tiles_per_img;
  hipStream_t s = (hipStream_t)stream;
  MM_PE_DISPATCH(nchw_ln_rows_bwd_kernel, dy, x, gamma, mu, rstd, dx, ws, C, HW, tiles_per_img, ntiles);
  return (int)hipGetLastError();
}

}  // extern "C"
