// Block-level glue of SS_Conv_SSM on gfx950: pure data movement, fused so that every tensor crosses HBM once.
//
// shuffle_residual: replaces  permute(0,2,3,1).contiguous() + cat + channel_shuffle(groups=2) + residual add
//   (MedMamba.py:354-357; channel_shuffle :308-320):
//     out[b,p,2i]   = left[b,i,p] + inp[b,p,2i]        left: conv-branch output, NCHW (b, C/2, P)
//     out[b,p,2i+1] = ssm[b,p,i]  + inp[b,p,2i+1]      ssm : SS2D-branch output, NHWC (b, P, C/2) or channel-first (b, C/2, P)
//   The NCHW->NHWC transpose of `left` goes through a 32x33 LDS tile; all global accesses are 128-B runs.
//   The backward is the same permutation read the other way (d_left, d_ssm from dout; d_inp = dout).
#include <cstdlib>
#include <stdlib.h>
#include "mm_common.h"
#include "medmamba_hip.h"

namespace {
// Ordering between the lanes of ONE wavefront through its own LDS region (one plane per wavefront kernels): LDS instructions of a
// wave execute in order, so a later ds_read sees an earlier ds_write of any lane of the same wave — all that is needed is that the
// compiler keeps the order.  A workgroup barrier here made the 4 independent waves of a workgroup wait for each other three times per plane.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

using namespace mm;

// bounds-checked dword access through a buffer descriptor (an offset of kOOB reads 0 / drops the store): see the depthwise kernels
__device__ __forceinline__ float ldb(rsrc_t r, int off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void stb(rsrc_t r, int off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}

// grid: (ceil(P/32), ceil(C2/32), B); block 256 = 8 rows x 32 lanes
template <bool SSM_CF>
__global__ __launch_bounds__(256) void shuffle_residual_fwd_kernel(const float* __restrict__ left, const float* __restrict__ ssm,
                                                                   const float* __restrict__ inp, float* __restrict__ out,
                                                                   const float* __restrict__ ssm_scale, int left_relu,
                                                                   const float* __restrict__ left_bias,
                                                                   int64_t ssm_sb, int64_t ssm_sd, int P, int C2) {
  __shared__ float tile[32][33];
  __shared__ float tile2[SSM_CF ? 32 : 1][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int p0 = blockIdx.x * 32, i0 = blockIdx.y * 32, b = blockIdx.z;
  const float* lb = left + (int64_t)b * C2 * P;
  const float sc = ssm_scale ? ssm_scale[b] : 1.0f;        // DropPath keep-mask / keep_prob of sample b
  const float lo = left_relu ? 0.0f : -__builtin_inff();   // trailing ReLU of the conv branch
  // load left[i0+r][p0+tx] (lanes along p)
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, p = p0 + tx;
    tile[r][tx] = (i < C2 && p < P) ? lb[(int64_t)i * P + p] + (left_bias ? left_bias[i] : 0.f) : 0.f;
    if constexpr (SSM_CF) tile2[r][tx] = (i < C2 && p < P) ? ssm[b * ssm_sb + i * ssm_sd + p] : 0.f;
  }
  __syncthreads();
  // out rows: position p0+r, channel pair i0+tx (lanes along i)
  const int64_t ob = (int64_t)b * P * (2 * C2);
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, i = i0 + tx;
    if (p < P && i < C2) {
      const float2 in2 = *reinterpret_cast<const float2*>(inp + ob + (int64_t)p * 2 * C2 + 2 * i);
      const float s = SSM_CF ? tile2[tx][r] : ssm[((int64_t)b * P + p) * C2 + i];
      *reinterpret_cast<float2*>(out + ob + (int64_t)p * 2 * C2 + 2 * i) =
          make_float2(fmaxf(tile[tx][r], lo) + in2.x, fmaf(s, sc, in2.y));
    }
  }
}

template <bool SSM_CF>
__global__ __launch_bounds__(256) void shuffle_residual_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dleft,
                                                                   float* __restrict__ dssm, const float* __restrict__ ssm_scale,
                                                                   const float* __restrict__ left_pre,
                                                                   const float* __restrict__ left_bias, int64_t dssm_sb,
                                                                   int64_t dssm_sd, int P, int C2) {
  __shared__ float tile[32][33];
  __shared__ float tile2[SSM_CF ? 32 : 1][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int p0 = blockIdx.x * 32, i0 = blockIdx.y * 32, b = blockIdx.z;
  const int64_t ob = (int64_t)b * P * (2 * C2);
  const float sc = ssm_scale ? ssm_scale[b] : 1.0f;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, i = i0 + tx;
    float2 g = make_float2(0.f, 0.f);
    if (p < P && i < C2) {
      g = *reinterpret_cast<const float2*>(dout + ob + (int64_t)p * 2 * C2 + 2 * i);
      g.y *= sc;
      if constexpr (!SSM_CF) dssm[((int64_t)b * P + p) * C2 + i] = g.y;
    }
    tile[r][tx] = g.x;       // [p][i]
    if constexpr (SSM_CF) tile2[r][tx] = g.y;
  }
  __syncthreads();
  float* lb = dleft + (int64_t)b * C2 * P;
  const float* lp = left_pre ? left_pre + (int64_t)b * C2 * P : nullptr;   // ReLU mask from the pre-activation
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, p = p0 + tx;
    if (i < C2 && p < P) {
      const bool on = lp ? lp[(int64_t)i * P + p] + (left_bias ? left_bias[i] : 0.f) > 0.0f : true;
      lb[(int64_t)i * P + p] = on ? tile[tx][r] : 0.0f;
      if constexpr (SSM_CF) dssm[b * dssm_sb + i * dssm_sd + p] = tile2[tx][r];
    }
  }
}
}  // namespace

extern "C" {

int mm_shuffle_residual_fwd(const float* left, const float* ssm, int64_t ssm_sb, int64_t ssm_sd, const float* inp, float* out,
                            const float* ssm_scale, int left_relu, const float* left_bias, int batch, int P, int C2,
                            int ssm_channel_first, void* stream) {
  if (ssm_sb == 0 && ssm_sd == 0) { ssm_sb = (int64_t)C2 * P; ssm_sd = P; }
  if (!left || !ssm || !inp || !out) return MM_ERR_NULL;
  if (batch <= 0 || P <= 0 || C2 <= 0 || batch > 65535) return MM_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(inp) | reinterpret_cast<uintptr_t>(out)) & 7) return MM_ERR_ALIGN;
  dim3 grid((P + 31) / 32, (C2 + 31) / 32, batch);
  if (ssm_channel_first) hipLaunchKernelGGL(shuffle_residual_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, left, ssm, inp, out, ssm_scale, left_relu, left_bias, ssm_sb, ssm_sd, P, C2);
  else hipLaunchKernelGGL(shuffle_residual_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, left, ssm, inp, out, ssm_scale, left_relu, left_bias, ssm_sb, ssm_sd, P, C2);
  return (int)hipGetLastError();
}

int mm_shuffle_residual_bwd(const float* dout, float* dleft, float* dssm, int64_t dssm_sb, int64_t dssm_sd,
                            const float* ssm_scale, const float* left_pre, const float* left_bias, int batch, int P, int C2,
                            int ssm_channel_first, void* stream) {
  if (dssm_sb == 0 && dssm_sd == 0) { dssm_sb = (int64_t)C2 * P; dssm_sd = P; }
  if (!dout || !dleft || !dssm) return MM_ERR_NULL;
  if (batch <= 0 || P <= 0 || C2 <= 0 || batch > 65535) return MM_ERR_SHAPE;
  if (reinterpret_cast<uintptr_t>(dout) & 7) return MM_ERR_ALIGN;
  dim3 grid((P + 31) / 32, (C2 + 31) / 32, batch);
  if (ssm_channel_first) hipLaunchKernelGGL(shuffle_residual_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, dout, dleft, dssm, ssm_scale, left_pre, left_bias, dssm_sb, dssm_sd, P, C2);
  else hipLaunchKernelGGL(shuffle_residual_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, dout, dleft, dssm, ssm_scale, left_pre, left_bias, dssm_sb, dssm_sd, P, C2);
  return (int)hipGetLastError();
}

}  // extern "C"

// =====================================================================================================
// SS2D in channel-first planes.  Between in_proj and out_proj every tensor lives as (batch, channel, H*W)
// planes, so the reference's NHWC<->NCHW permute copies (MedMamba.py:294, 299) and the transposes of the
// cross-scan / cross-merge (:256, :283-284) become per-plane operations with 128-B runs on both sides.
// =====================================================================================================
namespace {

// Plane loads of the depthwise kernels: a bounds-checked buffer load per element (an offset of kOOB reads 0: the zero border, the
// tail of a loop, a missing plane) — no branch around the load, so that the U loads of an unrolled chunk are ALL in flight before
// the first one is used.  (Round 4: the loops used to run `global_load_dword; s_waitcnt vmcnt(0)` per iteration — one load in flight
// per thread, ~5 KB per CU where ~40 KB are needed to cover HBM latency: the strip kernels ran at 3.1-3.3 TB/s; batched: fused
// backward 198.6 -> 140.6 us at 56x56 (4.4 TB/s), 93.0 -> 63.7 us at 28x28, forward 58.5 -> 52.5 / 31.5 -> 29.1 us.  The
// one-wavefront-per-plane kernels below keep their simple loops: with seven descriptors and 28 loads per lane up front they got
// SLOWER, 44.1 -> 53.2 us at 14x14 — many small waves already keep enough loads in flight there.  Nor are they load-bound at all:
// one b128 load per lane and operand plane (a 14x14 plane is 49 float4), parked in LDS in memory order, moved the fused backward
// 41.8 -> 41.2 us and the forward 22.1 -> 21.6 us — ~520 VALU / LDS instructions per plane and wave are what these kernels cost.)
constexpr int kDwU = 4;      // elements per thread and chunk

// ---- depthwise 3x3 conv + bias + SiLU (MedMamba.py:153-162, 295), writing the scan's two input orders --------
// x: planes (b, d) of H*W floats, batch stride x_sb, channel stride H*W.  out u2: (batch, 2, D, L):
// u2[b,0,d,h*W+w] = u2[b,1,d,w*H+h] = silu(conv(x)[b,d,h,w] + bias[d]).
// One workgroup per (plane, row strip): rows [r0, r0 + SH) of the plane plus a one-row halo sit in LDS with a zero
// border; the transposed copy goes through a second LDS tile.  Planes that fit the LDS budget whole are ONE strip
// (SH = H: the 56x56 stage and below); larger ones (96x96 of MedMamba-B at 384^2, anything bigger) are cut into strips
// of 32 rows, so there is no size limit and several workgroups fit a CU (MedMamba.py:288-305 runs at any resolution).
__global__ __launch_bounds__(256) void dwconv_silu_cross_fwd_kernel(const float* __restrict__ x, int64_t x_sb, int64_t x_sd,
                                                                    const float* __restrict__ wgt,
                                                                    const float* __restrict__ bias, float* __restrict__ u2,
                                                                    int64_t u_sb, int64_t u_sd, int D, int H, int W, int SH,
                                                                    int nstrips) {
  extern __shared__ float lds[];
  const int strip = blockIdx.x % nstrips, plane = blockIdx.x / nstrips;
  const int b = plane / D, d = plane % D;
  const int r0 = strip * SH, sh = min(SH, H - r0);           // rows of this strip
  const int WP = W + 2, tid = threadIdx.x, nt = blockDim.x;
  float* sx = lds;                         // (sh+2) x (W+2): rows r0-1 .. r0+sh, zero outside the plane
  float* so = lds + (SH + 2) * WP;         // sh x (W+1)
  const float* xp = x + (int64_t)b * x_sb + (int64_t)d * x_sd;
  const rsrc_t rx = make_rsrc(xp, (int64_t)H * W * 4);
  {
    DivMod dm(tid, nt, WP);
    const int N = (sh + 2) * WP;
    for (int i = tid; i < N; i += kDwU * nt) {
      float v[kDwU];
#pragma unroll
      for (int u = 0; u < kDwU; ++u) {
        const int hh = r0 + dm.q - 1, ww = dm.r - 1;
        v[u] = ldb(rx, (i + u * nt < N && hh >= 0 && hh < H && ww >= 0 && ww < W) ? (hh * W + ww) * 4 : kOOB);
        dm.next();
      }
#pragma unroll
      for (int u = 0; u < kDwU; ++u)
        if (i + u * nt < N) sx[i + u * nt] = v[u];
    }
  }
  float k[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) k[i] = wgt[d * 9 + i];
  const float bs = bias ? bias[d] : 0.f;
  __syncthreads();
  float* o0 = u2 + b * u_sb + d * u_sd;
  float* o1 = u2 + b * u_sb + (D + d) * u_sd;
  DivMod dw(tid, nt, W);
  for (int i = tid; i < sh * W; i += nt, dw.next()) {
    const int hl = dw.q, w = dw.r;
    const float* c = sx + hl * WP + w;     // top-left of the 3x3 window
    float p = bs;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) p = fmaf(c[kh * WP + kw], k[kh * 3 + kw], p);
    const float v = p * sigmoid_f(p);
    o0[(r0 + hl) * W + w] = v;
    so[hl * (W + 1) + w] = v;
  }
  __syncthreads();
  DivMod dh(tid, nt, sh);
  for (int i = tid; i < sh * W; i += nt, dh.next()) {      // i = w*sh + hl  (lanes along h: runs of sh floats in the column-major plane)
    const int w = dh.q, hl = dh.r;
    o1[w * H + r0 + hl] = so[hl * (W + 1) + w];
  }
}

// backward of the above: g = du2[b,0] + T(du2[b,1]);  dp = g * silu'(p);  dx = corr(dp, flipped k);
// per-(plane, strip) partial weight/bias gradients to ws[((b*D+d)*nstrips + strip)*10 + 0..8 | 9].
// du4 (optional): the scan's per-direction gradients of its input, 4*D planes per batch item; directions (0,1) read the
// row-major image and (2,3) the column-major one, so du2[b,j] += du4[b,2j] + du4[b,2j+1] is folded into the loads here
// (saves the pair-sum kernel and the read-modify-write of the projection GEMM that would otherwise accumulate into it).
// A strip needs dp on its rows and one row above / below, hence x on two rows above / below and g on one.
__global__ __launch_bounds__(256) void dwconv_silu_cross_bwd_kernel(const float* __restrict__ du2, int64_t g_sb, int64_t g_sd,
                                                                    const float* __restrict__ du4, int64_t e_sb, int64_t e_sd,
                                                                    const float* __restrict__ x, int64_t x_sb, int64_t x_sd,
                                                                    const float* __restrict__ wgt,
                                                                    const float* __restrict__ bias, float* __restrict__ dx,
                                                                    int64_t dx_sb, int64_t dx_sd, float* __restrict__ ws, int D,
                                                                    int H, int W, int SH, int nstrips) {
  extern __shared__ float lds[];
  const int strip = blockIdx.x % nstrips, plane = blockIdx.x / nstrips;
  const int b = plane / D, d = plane % D;
  const int r0 = strip * SH, sh = min(SH, H - r0);
  const int WP = W + 2, tid = threadIdx.x, nt = blockDim.x;
  const int ge0 = max(r0 - 1, 0), ge1 = min(r0 + sh + 1, H), gh = ge1 - ge0;   // rows that need dp: [ge0, ge1)
  float* sx = lds;                          // (sh+4) x (W+2): x rows r0-2 .. r0+sh+1, zero outside the plane
  float* sd = lds + (SH + 4) * WP;          // (sh+2) x (W+2): dp rows r0-1 .. r0+sh, zero outside the plane / strip halo
  float* st = sd + (SH + 2) * WP;           // (sh+2) x (W+1): transposed column-major gradient, rows r0-1 .. r0+sh
  __shared__ float red[10][4];
  const float* xp = x + (int64_t)b * x_sb + (int64_t)d * x_sd;
  const float* g0 = du2 + b * g_sb + d * g_sd;
  const float* g1 = du2 + b * g_sb + (D + d) * g_sd;
  const float* e0 = du4 ? du4 + b * e_sb + d * e_sd : nullptr;          // direction 0; direction k at e0 + k*D*e_sd
  const int64_t eD = (int64_t)D * e_sd;
  const int64_t pbytes = (int64_t)H * W * 4;
  const rsrc_t rx = make_rsrc(xp, pbytes), rg0 = make_rsrc(g0, pbytes), rg1 = make_rsrc(g1, pbytes);
  // the scan's four per-direction gradients (empty descriptors without du4: every load reads 0)
  const rsrc_t re0 = make_rsrc(e0, e0 ? pbytes : 0), re1 = make_rsrc(e0 ? e0 + eD : nullptr, e0 ? pbytes : 0),
               re2 = make_rsrc(e0 ? e0 + 2 * eD : nullptr, e0 ? pbytes : 0), re3 = make_rsrc(e0 ? e0 + 3 * eD : nullptr, e0 ? pbytes : 0);
  {
    DivMod dm(tid, nt, WP);
    const int N = (sh + 4) * WP;
    for (int i = tid; i < N; i += kDwU * nt) {
      float v[kDwU];
#pragma unroll
      for (int u = 0; u < kDwU; ++u) {
        const int hh = r0 + dm.q - 2, ww = dm.r - 1;
        v[u] = ldb(rx, (i + u * nt < N && hh >= 0 && hh < H && ww >= 0 && ww < W) ? (hh * W + ww) * 4 : kOOB);
        dm.next();
      }
#pragma unroll
      for (int u = 0; u < kDwU; ++u)
        if (i + u * nt < N) sx[i + u * nt] = v[u];
    }
  }
  for (int i = tid; i < (sh + 2) * WP; i += nt) sd[i] = 0.f;
  {
    DivMod dg(tid, nt, gh);
    const int N = gh * W;
    for (int i = tid; i < N; i += kDwU * nt) {   // i = w*gh + hg (lanes along h)
      float v[kDwU][3];
      int at[kDwU];
#pragma unroll
      for (int u = 0; u < kDwU; ++u) {
        const int w = dg.q, h = ge0 + dg.r;
        const int off = i + u * nt < N ? (w * H + h) * 4 : kOOB;
        v[u][0] = ldb(rg1, off); v[u][1] = ldb(re2, off); v[u][2] = ldb(re3, off);
        at[u] = (h - (r0 - 1)) * (W + 1) + w;
        dg.next();
      }
#pragma unroll
      for (int u = 0; u < kDwU; ++u)
        if (i + u * nt < N) st[at[u]] = v[u][0] + (v[u][1] + v[u][2]);
    }
  }
  float k[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) k[i] = wgt[d * 9 + i];
  const float bs = bias ? bias[d] : 0.f;
  __syncthreads();
  float acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = 0.f;
  DivMod dc(tid, nt, W);
  const int NG = gh * W;
  for (int i0 = tid; i0 < NG; i0 += kDwU * nt) {
    // the chunk's row-major gradients first (3 loads per element, all in flight), then the arithmetic
    float gin[kDwU];
    int hs[kDwU], wsv[kDwU];
#pragma unroll
    for (int u = 0; u < kDwU; ++u) {
      const int h = ge0 + dc.q, w = dc.r;
      const int off = i0 + u * nt < NG ? (h * W + w) * 4 : kOOB;
      gin[u] = ldb(rg0, off) + (ldb(re0, off) + ldb(re1, off));
      hs[u] = h; wsv[u] = w;
      dc.next();
    }
#pragma unroll
    for (int u = 0; u < kDwU; ++u) {
    if (i0 + u * nt < NG) {
    const int h = hs[u], w = wsv[u];
    const int hl = h - (r0 - 1);                        // row inside sd / st (0 .. sh+1)
    const float* c = sx + hl * WP + w;                 // sx row of h-1 is (h-1) - (r0-2) = hl
    float p = bs;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) p = fmaf(c[kh * WP + kw], k[kh * 3 + kw], p);
    const float sg = sigmoid_f(p);
    const float dp = (gin[u] + st[hl * (W + 1) + w]) * (sg * (1.f + p * (1.f - sg)));   // silu'(p) = s (1 + p (1 - s))
    sd[hl * WP + (w + 1)] = dp;
    if (h >= r0 && h < r0 + sh) {                       // parameter gradients: every row is owned by exactly one strip
      acc[9] += dp;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = fmaf(dp, c[kh * WP + kw], acc[kh * 3 + kw]);
    }
    }
    }
  }
  __syncthreads();
  float* dxp = dx + (int64_t)b * dx_sb + (int64_t)d * dx_sd;
  DivMod dxi(tid, nt, W);
  for (int i = tid; i < sh * W; i += nt, dxi.next()) {
    const int hl = dxi.q, w = dxi.r;
    // dx[h,w] = sum_{kh,kw} dp[h-kh+1, w-kw+1] * k[kh][kw]   (sd row of h+1-kh is hl + 2 - kh, column offset +1)
    float v = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) v = fmaf(sd[(hl + 2 - kh) * WP + (w - kw + 2)], k[kh * 3 + kw], v);
    dxp[(r0 + hl) * W + w] = v;
  }
  // strip reduction of the 10 partial sums: wave (DPP + shuffles), then across waves through LDS
  const int lane = tid & 63, wv = tid >> 6, nw = (nt + 63) >> 6;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    float v = group_sum<16>(acc[i]);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lane == 0) red[i][wv] = v;
  }
  __syncthreads();
  if (tid < 10) {
    float v = 0.f;
    for (int j = 0; j < nw; ++j) v += red[tid][j];
    ws[(int64_t)blockIdx.x * 10 + tid] = v;
  }
}

// ---- the same two kernels for planes of <= 256 positions (14x14, 7x7): one WAVEFRONT per plane, 4 planes per workgroup and
// iteration, each wave with its own LDS tiles (one workgroup of 128 threads per 196- or 49-element plane meant 24,576 /
// 49,152 workgroups with three barriers and a cross-wave reduction each).  Same arithmetic, same ws layout (nstrips = 1).
__global__ __launch_bounds__(256) void dwconv_silu_cross_fwd_small_kernel(const float* __restrict__ x, int64_t x_sb, int64_t x_sd,
                                                                          const float* __restrict__ wgt,
                                                                          const float* __restrict__ bias, float* __restrict__ u2,
                                                                          int64_t u_sb, int64_t u_sd, int D, int H, int W,
                                                                          int nplanes) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int WP = W + 2;
  const int per_wave = (H + 2) * WP + H * (W + 1);
  float* sx = lds + wv * per_wave;         // (H+2) x (W+2), zero border
  float* so = sx + (H + 2) * WP;           // H x (W+1)
  const int per_iter = gridDim.x * 4;
  const int niter = (nplanes + per_iter - 1) / per_iter;
  for (int it = 0; it < niter; ++it) {
    const int pl = it * per_iter + blockIdx.x * 4 + wv;
    const bool okp = pl < nplanes;
    const int b = okp ? pl / D : 0, d = okp ? pl % D : 0;
    const float* xp = x + (int64_t)b * x_sb + (int64_t)d * x_sd;
    {
      DivMod dm(lane, 64, WP);
      for (int i = lane; i < (H + 2) * WP; i += 64, dm.next()) {
        const int hh = dm.q - 1, ww = dm.r - 1;
        sx[i] = (okp && hh >= 0 && hh < H && ww >= 0 && ww < W) ? xp[hh * W + ww] : 0.f;
      }
    }
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = wgt[d * 9 + i];
    const float bs = bias ? bias[d] : 0.f;
    wave_lds_sync();
    float* o0 = u2 + b * u_sb + d * u_sd;
    float* o1 = u2 + b * u_sb + (D + d) * u_sd;
    for (DivMod dv(lane, 64, W); dv.q < H; dv.next()) {
      const int i = dv.q * W + dv.r, h = dv.q, w = dv.r;
      const float* c = sx + h * WP + w;
      float p = bs;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) p = fmaf(c[kh * WP + kw], k[kh * 3 + kw], p);
      const float v = p * sigmoid_f(p);
      if (okp) o0[i] = v;
      so[h * (W + 1) + w] = v;
    }
    wave_lds_sync();
    for (DivMod dv(lane, 64, H); dv.q < W; dv.next()) {      // i = w*H + h
      const int i = dv.q * H + dv.r, w = dv.q, h = dv.r;
      if (okp) o1[i] = so[h * (W + 1) + w];
    }
    wave_lds_sync();
  }
}

__global__ __launch_bounds__(256) void dwconv_silu_cross_bwd_small_kernel(const float* __restrict__ du2, int64_t g_sb, int64_t g_sd,
                                                                          const float* __restrict__ du4, int64_t e_sb, int64_t e_sd,
                                                                          const float* __restrict__ x, int64_t x_sb, int64_t x_sd,
                                                                          const float* __restrict__ wgt,
                                                                          const float* __restrict__ bias, float* __restrict__ dx,
                                                                          int64_t dx_sb, int64_t dx_sd, float* __restrict__ ws,
                                                                          int D, int H, int W, int nplanes) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int WP = W + 2;
  const int per_wave = (H + 2) * WP + (H + 2) * WP + H * (W + 1);
  float* sx = lds + wv * per_wave;          // (H+2) x (W+2): x with a zero border
  float* sd = sx + (H + 2) * WP;            // (H+2) x (W+2): dp with a zero border
  float* st = sd + (H + 2) * WP;            // H x (W+1): the column-major gradient, transposed
  const int per_iter = gridDim.x * 4;
  const int niter = (nplanes + per_iter - 1) / per_iter;
  for (int it = 0; it < niter; ++it) {
    const int pl = it * per_iter + blockIdx.x * 4 + wv;
    const bool okp = pl < nplanes;
    const int b = okp ? pl / D : 0, d = okp ? pl % D : 0;
    const float* xp = x + (int64_t)b * x_sb + (int64_t)d * x_sd;
    const float* g0 = du2 + b * g_sb + d * g_sd;
    const float* g1 = du2 + b * g_sb + (D + d) * g_sd;
    const float* e0 = du4 ? du4 + b * e_sb + d * e_sd : nullptr;
    const int64_t eD = (int64_t)D * e_sd;
    {
      DivMod dm(lane, 64, WP);
      for (int i = lane; i < (H + 2) * WP; i += 64, dm.next()) {
        const int hh = dm.q - 1, ww = dm.r - 1;
        sx[i] = (okp && hh >= 0 && hh < H && ww >= 0 && ww < W) ? xp[hh * W + ww] : 0.f;
        sd[i] = 0.f;
      }
    }
    for (DivMod dv(lane, 64, H); dv.q < W; dv.next()) {     // i = w*H + h (lanes along h)
      const int i = dv.q * H + dv.r, w = dv.q, h = dv.r;
      st[h * (W + 1) + w] = okp ? g1[i] + (e0 ? e0[2 * eD + i] + e0[3 * eD + i] : 0.f) : 0.f;
    }
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = wgt[d * 9 + i];
    const float bs = bias ? bias[d] : 0.f;
    wave_lds_sync();
    float acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.f;
    for (DivMod dv(lane, 64, W); dv.q < H; dv.next()) {
      const int i = dv.q * W + dv.r, h = dv.q, w = dv.r;
      const float* c = sx + h * WP + w;
      float p = bs;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) p = fmaf(c[kh * WP + kw], k[kh * 3 + kw], p);
      const float sg = sigmoid_f(p);
      const float ge = (okp && e0) ? e0[i] + e0[eD + i] : 0.f;
      const float dp = ((okp ? g0[i] : 0.f) + ge + st[h * (W + 1) + w]) * (sg * (1.f + p * (1.f - sg)));
      sd[(h + 1) * WP + (w + 1)] = dp;
      acc[9] += dp;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = fmaf(dp, c[kh * WP + kw], acc[kh * 3 + kw]);
    }
    wave_lds_sync();
    float* dxp = dx + (int64_t)b * dx_sb + (int64_t)d * dx_sd;
    for (DivMod dv(lane, 64, W); dv.q < H; dv.next()) {
      const int i = dv.q * W + dv.r, h = dv.q, w = dv.r;
      float v = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) v = fmaf(sd[(h + 2 - kh) * WP + (w - kw + 2)], k[kh * 3 + kw], v);
      if (okp) dxp[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      float v = group_sum<16>(acc[i]);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane == 0 && okp) ws[(int64_t)pl * 10 + i] = v;
    }
    wave_lds_sync();
  }
}

inline int dw_small_grid(int nplanes) { const int b = (nplanes + 15) / 16; return b < 1 ? 1 : b; }   // 4 planes x 4 iterations

// row-strip plan of the two kernels above: whole plane when the backward's three LDS tiles fit 32 KB, else 32-row strips
// (narrower strips for very wide planes so that a strip always fits)
struct DwPlan { int SH, nstrips; size_t lds_fwd, lds_bwd; };
inline DwPlan dw_plan(int H, int W) {
  auto need_bwd = [&](int sh) { return sizeof(float) * ((size_t)(sh + 4) * (W + 2) + (size_t)(sh + 2) * (W + 2) + (size_t)(sh + 2) * (W + 1)); };
  static const size_t whole_kb = [] { const char* e = getenv("MM_DW_WHOLE_KB"); return (size_t)(e ? atoi(e) : 32); }();   // tuning knob; 56x56 planes in two
  // strips (24 instead of 40 KB of LDS per workgroup: 6 per CU) run the fused backward in 199 instead of 222 us at B = 64
  int SH = H;
  // planes of <= 256 positions always run the one-wavefront-per-plane kernels, which write ONE row of partial sums per plane:
  // the plan (and with it mm_dwconv_silu_cross_strips, which sizes the caller's workspace) must say so whatever the knob is
  if ((int64_t)H * W > 256 && need_bwd(SH) > whole_kb * 1024) {
    SH = 32;
    while (SH > 1 && need_bwd(SH) > 96 * 1024) SH >>= 1;
    if (SH > H) SH = H;
  }
  DwPlan p;
  p.SH = SH; p.nstrips = (H + SH - 1) / SH;
  p.lds_fwd = sizeof(float) * ((size_t)(SH + 2) * (W + 2) + (size_t)SH * (W + 1));
  p.lds_bwd = need_bwd(SH);
  return p;
}

// ---- cross-merge (MedMamba.py:282-286, 298): m[b,d,h*W+w] = o[b,0,d,hw] + o[b,1,d,hw] + o[b,2,d,wh] + o[b,3,d,wh] ----
// o: (batch, 4, D, L) in position order (directions: row-major fwd/rev, column-major fwd/rev).  32x32 tiles per plane.
// grid: ceil(W/32) * ceil(H/32) * batch*D blocks (flattened)
__global__ __launch_bounds__(256) void cross_merge_fwd_kernel(const float* __restrict__ o, float* __restrict__ m, int64_t m_sb,
                                                              int64_t m_sd, int D, int H, int W) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int nbw = (W + 31) / 32, nbh = (H + 31) / 32;
  const int w0 = (blockIdx.x % nbw) * 32, h0 = ((blockIdx.x / nbw) % nbh) * 32, pl = blockIdx.x / (nbw * nbh);
  const int b = pl / D, d = pl % D;
  const int64_t L = (int64_t)H * W;
  const float* o0 = o + (((int64_t)b * 4 + 0) * D + d) * L;
  const float* o1 = o0 + (int64_t)D * L;
  const float* o2 = o1 + (int64_t)D * L;
  const float* o3 = o2 + (int64_t)D * L;
  // one descriptor per plane, no branch around a load: the 16 loads of a thread are in flight together (§4.4)
  const rsrc_t r0 = make_rsrc(o0, L * 4), r1 = make_rsrc(o1, L * 4), r2 = make_rsrc(o2, L * 4), r3 = make_rsrc(o3, L * 4);
  float c2[4], c3[4], a0[4], a1[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = ty + 8 * q;
    const int wc = w0 + r, hc = h0 + tx;     // rows w0+r of the column-major planes, lanes along h
    const int oc = (wc < W && hc < H) ? (wc * H + hc) * 4 : kOOB;
    c2[q] = ldb(r2, oc); c3[q] = ldb(r3, oc);
    const int hr = h0 + r, wr = w0 + tx;     // rows h0+r of the row-major planes, lanes along w
    const int orow = (hr < H && wr < W) ? (hr * W + wr) * 4 : kOOB;
    a0[q] = ldb(r0, orow); a1[q] = ldb(r1, orow);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) tile[ty + 8 * q][tx] = c2[q] + c3[q];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = ty + 8 * q, h = h0 + r, w = w0 + tx;
    if (h < H && w < W) m[b * m_sb + d * m_sd + (int64_t)h * W + w] = a0[q] + a1[q] + tile[tx][r];
  }
}

// plane transpose: dst[pl, w*H+h] = src[pl, h*W+w]; plane pl of batch b / channel d at b*sb + d*sd.
__global__ __launch_bounds__(256) void plane_transpose_kernel(const float* __restrict__ src, int64_t src_sb, int64_t src_sd,
                                                              float* __restrict__ dst, int64_t dst_sb, int64_t dst_sd, int D,
                                                              int H, int W) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int nbw = (W + 31) / 32, nbh = (H + 31) / 32;
  const int w0 = (blockIdx.x % nbw) * 32, h0 = ((blockIdx.x / nbw) % nbh) * 32, pl = blockIdx.x / (nbw * nbh);
  const int b = pl / D, d = pl % D;
  const float* s = src + (int64_t)b * src_sb + (int64_t)d * src_sd;
  float* t = dst + (int64_t)b * dst_sb + (int64_t)d * dst_sd;
  const rsrc_t rs = make_rsrc(s, (int64_t)H * W * 4);
  float v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int h = h0 + ty + 8 * q, w = w0 + tx;
    v[q] = ldb(rs, (h < H && w < W) ? (h * W + w) * 4 : kOOB);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) tile[ty + 8 * q][tx] = v[q];
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int w = w0 + r, h = h0 + tx;
    if (w < W && h < H) t[(int64_t)w * H + h] = tile[tx][r];
  }
}

// ---- small planes (H*W <= 256: the 14x14 and 7x7 stages): one WAVEFRONT per plane, 4 planes per workgroup and iteration.  The
// 32x32-tile kernels above spend a 256-thread workgroup on 196 or 49 elements (24,576 / 49,152 workgroups per call: 33 / 60 us
// for the merge, 30 / 56 us for the transpose at B = 64).  Lane l owns elements l, l+64, l+128, l+192 of the plane; the
// transposed operand goes through a 256-float LDS row of the wave.
// MERGE: m[pl, h*W+w] = o0 + o1 (row-major planes) + (o2 + o3)[w*H+h];   else: dst[pl, w*H+h] = src[pl, h*W+w]
template <bool MERGE>
__global__ __launch_bounds__(256) void plane_small_kernel(const float* __restrict__ src, int64_t src_sb, int64_t src_sd,
                                                          float* __restrict__ dst, int64_t dst_sb, int64_t dst_sd, int D, int H,
                                                          int W, int nplanes) {
  __shared__ float tile[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int L = H * W;
  int tr[4];                                   // transposed index of element lane + 64 j
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = lane + 64 * j, h = i / W, w = i - h * W;
    tr[j] = i < L ? w * H + h : 0;
  }
  const int per_iter = gridDim.x * 4;
  const int niter = (nplanes + per_iter - 1) / per_iter;
  for (int it = 0; it < niter; ++it) {
    const int pl = it * per_iter + blockIdx.x * 4 + wv;
    const bool okp = pl < nplanes;
    const int b = okp ? pl / D : 0, d = okp ? pl % D : 0;
    float* t = dst + (int64_t)b * dst_sb + (int64_t)d * dst_sd;
    if constexpr (MERGE) {
      const float* o0 = src + (((int64_t)b * 4 + 0) * D + d) * L;      // src = out4 (batch, 4, D, L) contiguous
      const float* o1 = o0 + (int64_t)D * L;
      const float* o2 = o1 + (int64_t)D * L;
      const float* o3 = o2 + (int64_t)D * L;
      float a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        const bool in = okp && i < L;
        tile[wv][i] = in ? o2[i] + o3[i] : 0.f;
        a[j] = in ? o0[i] + o1[i] : 0.f;
      }
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        if (okp && i < L) t[i] = a[j] + tile[wv][tr[j]];
      }
    } else {
      const float* sp = src + (int64_t)b * src_sb + (int64_t)d * src_sd;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        if (okp && i < L) tile[wv][tr[j]] = sp[i];
      }
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        if (okp && i < L) t[i] = tile[wv][i];
      }
    }
    wave_lds_sync();                           // the tile is rewritten by the next iteration
  }
}

inline int small_plane_grid(int nplanes) {     // 4 planes per workgroup and iteration, up to 8 iterations
  const int b = (nplanes + 31) / 32;
  return b < 1 ? 1 : b;
}

// ---- LayerNorm over channels (out_norm, MedMamba.py:300) + gate y*silu(z) (:301), channel-first ----------------
// m, y: (batch, D, L) contiguous; z: planes with batch stride z_sb.  Thread layout: PW consecutive positions x TPP
// channel chunks per wave (TPP = 64/PW lanes share a position and split D); statistics mu/rstd: (batch, L).
// sum over the W lanes that share lane / W (W = 1..64, the low lane bits), every lane gets the sum: DPP inside the 16-lane
// rows, v_permlane16/32_swap across them — no ds_bpermute (the generic __shfl_xor) on these hot paths
template <int W>
__device__ __forceinline__ float low_sum(float v) {
  v = mm::group_sum<(W < 16 ? W : 16)>(v);
  if constexpr (W >= 32) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    const unsigned a = r[0], b = r[1];
    v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
  }
  if constexpr (W >= 64) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    const unsigned a = r[0], b = r[1];
    v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
  }
  return v;
}

// Channel rows of the LayerNorm + gate kernels through buffer descriptors (Round 4): the CPL loads of a lane used to sit behind
// `in ? p[d * sd] : 0` — a branch per load and `s_waitcnt vmcnt(0)` before the next one, 3-4 loads in flight per lane where the
// latency of HBM wants dozens (ln_gate_bwdc<32,8,24> alone: 1.7 TB/s).  Here: one descriptor per step k that starts at the first
// channel row of the wave's step (a wave-uniform 64-bit base: no 32-bit limit on D * channel stride), a per-lane byte offset
// (row inside the step, position) that is kOOB for lanes without work — no control flow, every load issued up front.
// (the host checks (rows - 1) * sd + L < 2^29 elements: mm_ln_gate_fwd / _bwd)
__device__ __forceinline__ rsrc_t rows_rsrc(const float* base, int d0, int64_t sd, int rows, int L) {
  return make_rsrc(base + (int64_t)d0 * sd, ((int64_t)(rows - 1) * sd + L) * 4);
}

template <int PW>
__device__ __forceinline__ float pos_sum(float v) {   // sum over the 64/PW lanes that share a position
#pragma unroll
  for (int s = PW; s < 64; s <<= 1) v += __shfl_xor(v, s);
  return v;
}

template <int PW>
__global__ __launch_bounds__(256) void ln_gate_fwd_kernel(const float* __restrict__ m, int64_t m_sb, int64_t m_sd,
                                                          const float* __restrict__ z, int64_t z_sb, int64_t z_sd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float* __restrict__ y, int64_t y_sb, int64_t y_sd,
                                                          float* __restrict__ mu_out, float* __restrict__ rstd_out, int D, int L,
                                                          int npos_blocks) {
  constexpr int TPP = 64 / PW;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.x / npos_blocks, pb = blockIdx.x % npos_blocks;
  const int p = (pb * 4 + wv) * PW + (lane % PW);
  const int ck = lane / PW;                               // channel chunk of this lane
  const bool ok = p < L;
  const float* mp = m + (int64_t)b * m_sb + p;
  const float* zp = z + (int64_t)b * z_sb + p;
  const float shift = ok ? mp[0] : 0.f;                   // shifted one-pass variance (shift = channel 0)
  float s1 = 0.f, s2 = 0.f;
  for (int d = ck; d < D; d += TPP) {
    const float v = ok ? mp[d * m_sd] - shift : 0.f;
    s1 += v;
    s2 = fmaf(v, v, s2);
  }
  s1 = pos_sum<PW>(s1);
  s2 = pos_sum<PW>(s2);
  const float mean_s = s1 / D;
  const float var = fmaxf(s2 / D - mean_s * mean_s, 0.f);
  const float mu = mean_s + shift, rstd = __builtin_amdgcn_rsqf(var + eps);
  if (ok && ck == 0) {
    mu_out[(int64_t)b * L + p] = mu;
    rstd_out[(int64_t)b * L + p] = rstd;
  }
  if (!ok) return;
  float* yp = y + (int64_t)b * y_sb + p;
  for (int d = ck; d < D; d += TPP) {
    const float n = (mp[d * m_sd] - mu) * rstd * gamma[d] + beta[d];
    const float zz = zp[d * z_sd];
    yp[d * y_sd] = n * (zz * sigmoid_f(zz));
  }
}

// backward: dm (written with batch stride dm_sb), dz (batch stride dz_sb), and per-wave partial sums of
// dgamma / dbeta to ws[(wave_global) * 2 * D + {0: dgamma, D: dbeta} + d]  (summed by the caller).
template <int PW>
__global__ __launch_bounds__(256) void ln_gate_bwd_kernel(const float* __restrict__ dy, int64_t g_sb, int64_t g_sd,
                                                          const float* __restrict__ m, int64_t m_sb, int64_t m_sd,
                                                          const float* __restrict__ z, int64_t z_sb, int64_t z_sd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mu_in, const float* __restrict__ rstd_in,
                                                          float* __restrict__ dm, int64_t dm_sb, int64_t dm_sd,
                                                          float* __restrict__ dz, int64_t dz_sb, int64_t dz_sd,
                                                          float* __restrict__ ws, int D, int L, int npos_blocks) {
  constexpr int TPP = 64 / PW;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.x / npos_blocks, pb = blockIdx.x % npos_blocks;
  const int p = (pb * 4 + wv) * PW + (lane % PW);
  const int ck = lane / PW;
  const bool ok = p < L;
  const float* mp = m + (int64_t)b * m_sb + p;
  const float* gp = dy + (int64_t)b * g_sb + p;
  const float* zp = z + (int64_t)b * z_sb + p;
  const float mu = ok ? mu_in[(int64_t)b * L + p] : 0.f, rstd = ok ? rstd_in[(int64_t)b * L + p] : 0.f;
  float c1 = 0.f, c2 = 0.f;
  for (int d = ck; d < D; d += TPP) {
    if (ok) {
      const float zz = zp[d * z_sd];
      const float dn = gp[d * g_sd] * (zz * sigmoid_f(zz)) * gamma[d];
      const float xh = (mp[d * m_sd] - mu) * rstd;
      c1 += dn;
      c2 = fmaf(dn, xh, c2);
    }
  }
  c1 = pos_sum<PW>(c1) / D;
  c2 = pos_sum<PW>(c2) / D;
  float* dmp = dm + (int64_t)b * dm_sb + p;
  float* dzp = dz + (int64_t)b * dz_sb + p;
  extern __shared__ float sred[];                       // [4 waves][2*D]: dgamma | dbeta, one row per wave (plain stores: every
                                                        // (wave, d) is written exactly once), summed in wave order below — no LDS
                                                        // float atomics, so the result does not depend on the arrival order
  for (int d0 = 0; d0 < D; d0 += TPP) {
    const int d = d0 + ck;
    float pg = 0.f, pb_ = 0.f;
    if (ok && d < D) {
      const float zz = zp[d * z_sd], s = sigmoid_f(zz), sz = zz * s;
      const float g = gp[d * g_sd];
      const float xh = (mp[d * m_sd] - mu) * rstd;
      const float n = xh * gamma[d] + beta[d];
      const float dn = g * sz;
      dzp[d * dz_sd] = g * n * (s * (1.f + zz * (1.f - s)));
      dmp[d * dm_sd] = rstd * (dn * gamma[d] - c1 - xh * c2);
      pg = dn * xh;
      pb_ = dn;
    }
    // sum over the PW positions of this wave that share channel d (lanes with equal ck): xor over the low bits
    pg = low_sum<PW>(pg);
    pb_ = low_sum<PW>(pb_);
    if ((lane % PW) == 0 && d < D) {
      sred[wv * 2 * D + d] = pg;
      sred[wv * 2 * D + D + d] = pb_;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += blockDim.x)       // one row per workgroup
    ws[(int64_t)blockIdx.x * 2 * D + i] = (sred[i] + sred[2 * D + i]) + (sred[4 * D + i] + sred[6 * D + i]);
}

inline int pick_pw(int batch, int L) {   // positions per wave: fewer when there are few positions (small images)
  const long npos = (long)batch * L;
  if (npos >= 64 * 1024) return 64;
  if (npos >= 16 * 1024) return 16;
  return 4;
}

// ---- single-pass variants: the CPL channels a lane owns stay in registers between the statistics and the output pass,
// so every operand crosses HBM once (the two-pass kernels above re-read m / dy / z: their working set, 3 x 77 MB at the
// 56x56 stage, does not survive in L2).  TPP = 64/PW lanes share a position; CPL >= ceil(D / TPP).
template <int PW, int CPL>
__global__ __launch_bounds__(256) void ln_gate_fwd1_kernel(const float* __restrict__ m, int64_t m_sb, int64_t m_sd,
                                                           const float* __restrict__ z, int64_t z_sb, int64_t z_sd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, float* __restrict__ y, int64_t y_sb, int64_t y_sd,
                                                           float* __restrict__ mu_out, float* __restrict__ rstd_out, int D,
                                                           int L, int npos_blocks) {
  constexpr int TPP = 64 / PW;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.x / npos_blocks, pb = blockIdx.x % npos_blocks;
  const int p = (pb * 4 + wv) * PW + (lane % PW);
  const int ck = lane / PW;
  const bool ok = p < L;
  const float* mb = m + (int64_t)b * m_sb;
  const float* zb = z + (int64_t)b * z_sb;
  const int om = ok ? (ck * (int)m_sd + p) * 4 : kOOB, oz = ok ? (ck * (int)z_sd + p) * 4 : kOOB;
  float mv[CPL], zv[CPL], s1 = 0.f;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const bool in = ck + k * TPP < D;
    mv[k] = ldb(rows_rsrc(mb, k * TPP, m_sd, TPP, L), in ? om : kOOB);
    zv[k] = ldb(rows_rsrc(zb, k * TPP, z_sd, TPP, L), in ? oz : kOOB);
  }
#pragma unroll
  for (int k = 0; k < CPL; ++k) s1 += mv[k];
  const float mu = pos_sum<PW>(s1) / D;
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const float dv = (ck + k * TPP < D) ? mv[k] - mu : 0.f;
    s2 = fmaf(dv, dv, s2);
  }
  const float rstd = __builtin_amdgcn_rsqf(pos_sum<PW>(s2) / D + eps);
  if (ok && ck == 0) {
    mu_out[(int64_t)b * L + p] = mu;
    rstd_out[(int64_t)b * L + p] = rstd;
  }
  float* yb = y + (int64_t)b * y_sb;
  const int oy = ok ? (ck * (int)y_sd + p) * 4 : kOOB;
  const rsrc_t rgam = make_rsrc(gamma, (int64_t)D * 4), rbet = make_rsrc(beta, (int64_t)D * 4);
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int d = ck + k * TPP;                                     // d >= D: gamma / beta read 0, the store is dropped
    const float n = (mv[k] - mu) * rstd * ldb(rgam, d * 4) + ldb(rbet, d * 4);
    const float zz = zv[k];
    stb(rows_rsrc(yb, k * TPP, y_sd, TPP, L), d < D ? oy : kOOB, n * (zz * sigmoid_f(zz)));
  }
}

template <int PW, int CPL>
__global__ __launch_bounds__(256) void ln_gate_bwd1_kernel(const float* __restrict__ dy, int64_t g_sb, int64_t g_sd,
                                                           const float* __restrict__ m, int64_t m_sb, int64_t m_sd,
                                                           const float* __restrict__ z, int64_t z_sb, int64_t z_sd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mu_in, const float* __restrict__ rstd_in,
                                                           float* __restrict__ dm, int64_t dm_sb, int64_t dm_sd,
                                                           float* __restrict__ dz, int64_t dz_sb, int64_t dz_sd,
                                                           float* __restrict__ ws, int D, int L, int npos_blocks) {
  constexpr int TPP = 64 / PW;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.x / npos_blocks, pb = blockIdx.x % npos_blocks;
  const int p = (pb * 4 + wv) * PW + (lane % PW);
  const int ck = lane / PW;
  const bool ok = p < L;
  const float* mp = m + (int64_t)b * m_sb + p;
  const float* gp = dy + (int64_t)b * g_sb + p;
  const float* zp = z + (int64_t)b * z_sb + p;
  const float mu = ok ? mu_in[(int64_t)b * L + p] : 0.f, rstd = ok ? rstd_in[(int64_t)b * L + p] : 0.f;
  float gv[CPL], zv[CPL], xh[CPL], c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int d = ck + k * TPP;
    const bool in = ok && d < D;
    zv[k] = in ? zp[d * z_sd] : 0.f;
    gv[k] = in ? gp[d * g_sd] : 0.f;
    xh[k] = in ? (mp[d * m_sd] - mu) * rstd : 0.f;
    const float dn = in ? gv[k] * (zv[k] * sigmoid_f(zv[k])) * gamma[d] : 0.f;
    c1 += dn;
    c2 = fmaf(dn, xh[k], c2);
  }
  c1 = pos_sum<PW>(c1) / D;
  c2 = pos_sum<PW>(c2) / D;
  float* dmp = dm + (int64_t)b * dm_sb + p;
  float* dzp = dz + (int64_t)b * dz_sb + p;
  extern __shared__ float sred[];                       // [4 waves][2*D]: dgamma | dbeta, one row per wave (plain stores: every
                                                        // (wave, d) is written exactly once), summed in wave order below — no LDS
                                                        // float atomics, so the result does not depend on the arrival order
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int d = ck + k * TPP;
    float pg = 0.f, pb_ = 0.f;
    if (ok && d < D) {
      const float zz = zv[k], s_ = sigmoid_f(zz), sz = zz * s_;
      const float gm = gamma[d];
      const float n = xh[k] * gm + beta[d];
      const float dn = gv[k] * sz;
      dzp[d * dz_sd] = gv[k] * n * (s_ * (1.f + zz * (1.f - s_)));
      dmp[d * dm_sd] = rstd * (dn * gm - c1 - xh[k] * c2);
      pg = dn * xh[k];
      pb_ = dn;
    }
    // sum over the PW positions of this wave that share channel d (lanes with equal ck): xor over the low bits
    pg = low_sum<PW>(pg);
    pb_ = low_sum<PW>(pb_);
    if ((lane % PW) == 0 && d < D) {
      sred[wv * 2 * D + d] = pg;
      sred[wv * 2 * D + D + d] = pb_;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += blockDim.x)       // one row per workgroup
    ws[(int64_t)blockIdx.x * 2 * D + i] = (sred[i] + sred[2 * D + i]) + (sred[4 * D + i] + sred[6 * D + i]);
}

// ---- cooperative single-pass variants for wide channel counts: the D channels of a position are split over ALL lanes of the
// workgroup that share it (64/PW in each of the NW waves), the statistics meet in LDS.  With 16 lanes per position inside ONE
// wave (ln_gate_*1_kernel<4, ...>, the D > 128 plan so far) a wave-wide load touched only 4 consecutive positions = 16 B of each
// channel row: 1.4-1.6 TB/s at the 14x14 stage.  Here PW = 16 positions are contiguous per channel (64-B runs) at the same
// registers per lane (CPL = D / (NW * 64 / PW)).
template <int PW, int NW, int CPL>
__global__ __launch_bounds__(NW * 64) void ln_gate_fwdc_kernel(const float* __restrict__ m, int64_t m_sb, int64_t m_sd,
                                                               const float* __restrict__ z, int64_t z_sb, int64_t z_sd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, float* __restrict__ y, int64_t y_sb, int64_t y_sd,
                                                               float* __restrict__ mu_out, float* __restrict__ rstd_out, int D,
                                                               int L, int npos_blocks) {
  constexpr int LPW = 64 / PW, TPP = LPW * NW;
  __shared__ float sstat[2][NW][PW];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / npos_blocks, pb = blockIdx.x % npos_blocks;
  const int pl = lane % PW, p = pb * PW + pl;
  const int lg = lane / PW, ck = wv * LPW + lg;
  const bool ok = p < L;
  const float* mb = m + (int64_t)b * m_sb;
  const float* zb = z + (int64_t)b * z_sb;
  const int om = ok ? (lg * (int)m_sd + p) * 4 : kOOB, oz = ok ? (lg * (int)z_sd + p) * 4 : kOOB;
  float mv[CPL], zv[CPL], s1 = 0.f;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int d0 = wv * LPW + k * TPP;                        // first channel row of this wave's step k (wave-uniform)
    const bool in = d0 + lg < D;
    mv[k] = ldb(rows_rsrc(mb, d0, m_sd, LPW, L), in ? om : kOOB);
    zv[k] = ldb(rows_rsrc(zb, d0, z_sd, LPW, L), in ? oz : kOOB);
  }
#pragma unroll
  for (int k = 0; k < CPL; ++k) s1 += mv[k];
  s1 = pos_sum<PW>(s1);
  if (lane < PW) sstat[0][wv][pl] = s1;
  __syncthreads();
  float mu = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) mu += sstat[0][w][pl];
  mu /= D;
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const float dv = (ck + k * TPP < D) ? mv[k] - mu : 0.f;
    s2 = fmaf(dv, dv, s2);
  }
  s2 = pos_sum<PW>(s2);
  if (lane < PW) sstat[1][wv][pl] = s2;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) var += sstat[1][w][pl];
  const float rstd = __builtin_amdgcn_rsqf(var / D + eps);
  if (ok && ck == 0) {
    mu_out[(int64_t)b * L + p] = mu;
    rstd_out[(int64_t)b * L + p] = rstd;
  }
  float* yb = y + (int64_t)b * y_sb;
  const int oy = ok ? (lg * (int)y_sd + p) * 4 : kOOB;
  const rsrc_t rgam = make_rsrc(gamma, (int64_t)D * 4), rbet = make_rsrc(beta, (int64_t)D * 4);
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int d0 = wv * LPW + k * TPP, d = d0 + lg;          // d >= D: gamma / beta read 0, the store is dropped
    const float n = (mv[k] - mu) * rstd * ldb(rgam, d * 4) + ldb(rbet, d * 4);
    stb(rows_rsrc(yb, d0, y_sd, LPW, L), d < D ? oy : kOOB, n * (zv[k] * sigmoid_f(zv[k])));
  }
}

template <int PW, int NW, int CPL>
__global__ __launch_bounds__(NW * 64) void ln_gate_bwdc_kernel(const float* __restrict__ dy, int64_t g_sb, int64_t g_sd,
                                                               const float* __restrict__ m, int64_t m_sb, int64_t m_sd,
                                                               const float* __restrict__ z, int64_t z_sb, int64_t z_sd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ mu_in, const float* __restrict__ rstd_in,
                                                               float* __restrict__ dm, int64_t dm_sb, int64_t dm_sd,
                                                               float* __restrict__ dz, int64_t dz_sb, int64_t dz_sd,
                                                               float* __restrict__ ws, int D, int L, int npos_blocks) {
  constexpr int LPW = 64 / PW, TPP = LPW * NW;
  constexpr int NGB = (2 * CPL * TPP + NW * 64 - 1) / (NW * 64);   // gamma | beta values a thread carries into the LDS
  __shared__ float sstat[2][NW][PW];
  extern __shared__ float sred[];                       // [2*D]: dgamma | dbeta of this workgroup, then [2*D]: gamma | beta
  float* sgb = sred + 2 * D;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / npos_blocks, pb = blockIdx.x % npos_blocks;
  const int pl = lane % PW, p = pb * PW + pl;
  const int lg = lane / PW, ck = wv * LPW + lg;
  const bool ok = p < L;
  const float* mb = m + (int64_t)b * m_sb;
  const float* gb = dy + (int64_t)b * g_sb;
  const float* zb = z + (int64_t)b * z_sb;
  const rsrc_t rst = make_rsrc(mu_in + (int64_t)b * L, (int64_t)L * 4), rrs = make_rsrc(rstd_in + (int64_t)b * L, (int64_t)L * 4);
  const float mu = ldb(rst, ok ? p * 4 : kOOB), rstd = ldb(rrs, ok ? p * 4 : kOOB);
  // gamma | beta of all channels once per workgroup (an offset beyond D * 4 — also a negative one — reads 0)
  const rsrc_t rgam = make_rsrc(gamma, (int64_t)D * 4), rbet = make_rsrc(beta, (int64_t)D * 4);
  float gbv[NGB];
#pragma unroll
  for (int j = 0; j < NGB; ++j) {
    const int i = threadIdx.x + j * NW * 64;
    gbv[j] = ldb(rgam, i * 4) + ldb(rbet, (i - D) * 4);
  }
  const int om = ok ? (lg * (int)m_sd + p) * 4 : kOOB, og = ok ? (lg * (int)g_sd + p) * 4 : kOOB,
            oz = ok ? (lg * (int)z_sd + p) * 4 : kOOB;
  float gv[CPL], zv[CPL], xh[CPL], c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int d0 = wv * LPW + k * TPP;                        // first channel row of this wave's step k (wave-uniform)
    const bool in = d0 + lg < D;
    zv[k] = ldb(rows_rsrc(zb, d0, z_sd, LPW, L), in ? oz : kOOB);
    gv[k] = ldb(rows_rsrc(gb, d0, g_sd, LPW, L), in ? og : kOOB);
    xh[k] = ldb(rows_rsrc(mb, d0, m_sd, LPW, L), in ? om : kOOB);
  }
#pragma unroll
  for (int j = 0; j < NGB; ++j) {
    const int i = threadIdx.x + j * NW * 64;
    if (i < 2 * D) { sgb[i] = gbv[j]; sred[i] = 0.f; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int d = ck + k * TPP;
    const bool in = ok && d < D;
    xh[k] = in ? (xh[k] - mu) * rstd : 0.f;
    const float dn = gv[k] * (zv[k] * sigmoid_f(zv[k])) * sgb[d < D ? d : 0];       // gv = 0 where there is no work
    c1 += dn;
    c2 = fmaf(dn, xh[k], c2);
  }
  c1 = pos_sum<PW>(c1);
  c2 = pos_sum<PW>(c2);
  if (lane < PW) { sstat[0][wv][pl] = c1; sstat[1][wv][pl] = c2; }
  __syncthreads();
  c1 = 0.f; c2 = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) { c1 += sstat[0][w][pl]; c2 += sstat[1][w][pl]; }
  c1 /= D; c2 /= D;
  float* dmb = dm + (int64_t)b * dm_sb;
  float* dzb = dz + (int64_t)b * dz_sb;
  const int odm = ok ? (lg * (int)dm_sd + p) * 4 : kOOB, odz = ok ? (lg * (int)dz_sd + p) * 4 : kOOB;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int d0 = wv * LPW + k * TPP, d = d0 + lg, dc = d < D ? d : 0;
    const bool in = ok && d < D;
    const float zz = zv[k], s_ = sigmoid_f(zz), sz = zz * s_;
    const float gm = sgb[dc];
    const float n = xh[k] * gm + sgb[D + dc];
    const float dn = gv[k] * sz;                          // 0 where there is no work: gv = 0
    stb(rows_rsrc(dzb, d0, dz_sd, LPW, L), d < D ? odz : kOOB, gv[k] * n * (s_ * (1.f + zz * (1.f - s_))));
    stb(rows_rsrc(dmb, d0, dm_sd, LPW, L), d < D ? odm : kOOB, rstd * (dn * gm - c1 - xh[k] * c2));
    // sum over the PW positions of this wave that share channel d (lanes with equal lane / PW): xor over the low bits
    const float pg = low_sum<PW>(in ? dn * xh[k] : 0.f);
    const float pb_ = low_sum<PW>(in ? dn : 0.f);
    if (pl == 0 && d < D) {          // every channel belongs to exactly one (wave, lane group)
      sred[d] = pg;
      sred[D + d] = pb_;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += NW * 64) ws[(int64_t)blockIdx.x * 2 * D + i] = sred[i];   // one row per workgroup
}

// plan of the LayerNorm + gate kernels: positions per wave and, for the single-pass kernels, channels per lane
// (cpl = 0: D is too wide for the register-resident form -> two-pass kernels; nw > 0: cooperative kernel with nw waves)
struct LnPlan { int pw, cpl, nw, ppb; };     // ppb = positions per workgroup
inline LnPlan plan_ln(int batch, int D, int L, bool bwd) {
  LnPlan pl;
  // the backward keeps three values per channel: 48 channels per lane would be 256 VGPRs (1 wave per SIMD)
  static const int two_pass = [] { const char* e = getenv("MM_LN_TWO_PASS"); return e ? atoi(e) : 0; }();   // tuning knob: 1 = fwd, 2 = bwd, 3 = both
  static const int coop = [] { const char* e = getenv("MM_LN_COOP"); return e ? atoi(e) : 1; }();          // 0: never the cooperative kernels
  if (D > 128 && coop && !(two_pass & (bwd ? 2 : 1))) {
    // 16 positions per workgroup (64-B runs per channel row); lanes per position: 16 (4 waves) or 32 (8 waves), the fewest
    // that keep the channels of a lane within 32 (forward: two values per channel) / 24 (backward: three)
    const int cap = bwd ? 24 : 32;
    const int lpp = (D + 15) / 16 <= cap ? 16 : 32;
    const int need = (D + lpp - 1) / lpp;
    if (need <= cap) {
      // 16 lanes per position: 32 positions per workgroup (128-B runs, 8 waves) — measured (tools/bench_ln_gate.py, B = 64, us):
      // forward D = 192 31.6 -> 28.0, D = 384 24.8 -> 20.3; the backward did not gain while its tiles were cut per batch item
      // (61.4 -> 64.9, 55.8 -> 53.9) and does since the channel-major row is tiled as a whole (mm_ln_gate_bwd): D = 384 46.7 ->
      // 41.7, D = 192 56.4 -> 55.3 — and it leaves half as many partial rows to sum
      static const int pw32 = [] { const char* e = getenv("MM_LN_PW32"); return e ? atoi(e) : 1; }();
      const bool wide = lpp == 16 && (bwd ? pw32 != 3 && pw32 != 0 : pw32 != 0);   // MM_LN_PW32: 0 = never, 3 = forward only (A/B)
      pl.pw = wide ? 32 : 16; pl.nw = wide ? 8 : lpp / 4; pl.ppb = pl.pw;
      pl.cpl = (wide && need <= 12) ? 12 : need <= 16 ? 16 : need <= 24 ? 24 : need <= 32 ? 32 : 48;   // 12: D = 192 without idle slots
      return pl;
    }
  }
  if (D <= 128 && bwd && coop && !(two_pass & 2) && (long)batch * L >= 16 * 1024) {
    // backward for narrow D on many positions (the 56x56 / 96x96 stages): 64 positions per workgroup (256-B runs), one lane per
    // position in each of 4 waves, D/4 channels per lane — single pass instead of the two-pass kernel (S, B = 64, D = 96: 178 -> 163 us)
    pl.pw = 64; pl.nw = 4; pl.ppb = 64;
    const int need4 = (D + 3) / 4;
    pl.cpl = need4 <= 16 ? 16 : need4 <= 24 ? 24 : 32;
    return pl;
  }
  const int tpp = D <= 128 ? 4 : 16;
  const int need = (D + tpp - 1) / tpp;
  // measured (tools/bench_ln_gate.py, B = 64): forward 86 -> 48 / 39 -> 36 / 40 -> 31 / 38 -> 27 us for the four S stages;
  // backward only wins where 16 lanes share a position and still own <= 24 channels (D = 384: 80 -> 55 us) — with 4
  // lanes per position it writes 4x the partial-sum rows (165 -> 158 us at D = 96, 60 -> 70 us at D = 192)
  static const int bwd_all = [] { const char* e = getenv("MM_LN_BWD1_ALL"); return e ? atoi(e) : 0; }();
  const bool fits = bwd ? ((bwd_all ? true : (tpp == 16 && need > 16)) && need <= 24) : need <= 48;
  pl.nw = 0;
  if (fits && !(two_pass & (bwd ? 2 : 1))) {
    pl.pw = 64 / tpp;
    pl.cpl = need <= 8 ? 8 : need <= 16 ? 16 : need <= 24 ? 24 : need <= 32 ? 32 : 48;
  } else {
    pl.pw = pick_pw(batch, L);
    pl.cpl = 0;
  }
  pl.ppb = 4 * pl.pw;
  return pl;
}

inline bool ln_offsets_fit(int batch, int L, std::initializer_list<int64_t> channel_strides) {
  for (int64_t sd : channel_strides)
    if (sd < 0 || 15 * sd + (int64_t)batch * L >= (1ll << 29)) return false;
  return true;
}
inline bool ln_flat() { static const bool on = [] { const char* e = getenv("MM_LN_FLAT"); return e ? atoi(e) != 0 : true; }(); return on; }   // A/B switch

#define MM_LN_ARGS_F m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, eps, y, y_sb, y_sd, mu, rstd, D, L, npb
int launch_ln_fwdc(const LnPlan& pl, dim3 grid, hipStream_t s, const float* m, int64_t m_sb, int64_t m_sd, const float* z,
                   int64_t z_sb, int64_t z_sd, const float* gamma, const float* beta, float eps, float* y, int64_t y_sb, int64_t y_sd,
                   float* mu, float* rstd, int D, int L, int npb) {
#define MM_LN_FWDC(NW_, CPL_) hipLaunchKernelGGL((ln_gate_fwdc_kernel<16, NW_, CPL_>), grid, dim3(NW_ * 64), 0, s, MM_LN_ARGS_F)
#define MM_LN_FWDC32(CPL_) hipLaunchKernelGGL((ln_gate_fwdc_kernel<32, 8, CPL_>), grid, dim3(512), 0, s, MM_LN_ARGS_F)
  if (pl.pw == 32) {
    switch (pl.cpl) {
      case 12: MM_LN_FWDC32(12); break;
      case 16: MM_LN_FWDC32(16); break;
      case 24: MM_LN_FWDC32(24); break;
      default: MM_LN_FWDC32(32); break;
    }
  } else if (pl.nw == 4) {
    switch (pl.cpl) {
      case 16: MM_LN_FWDC(4, 16); break;
      case 24: MM_LN_FWDC(4, 24); break;
      case 32: MM_LN_FWDC(4, 32); break;
      default: MM_LN_FWDC(4, 48); break;
    }
  } else {
    switch (pl.cpl) {
      case 16: MM_LN_FWDC(8, 16); break;
      case 24: MM_LN_FWDC(8, 24); break;
      case 32: MM_LN_FWDC(8, 32); break;
      default: MM_LN_FWDC(8, 48); break;
    }
  }
#undef MM_LN_FWDC
  return (int)hipGetLastError();
}
#undef MM_LN_ARGS_F

#define MM_LN_ARGS_B dy, dy_sb, dy_sd, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, mu, rstd, dm, dm_sb, dm_sd, dz, dz_sb, dz_sd, ws, D, L, npb
int launch_ln_bwdc(const LnPlan& pl, dim3 grid, hipStream_t s, const float* dy, int64_t dy_sb, int64_t dy_sd, const float* m,
                   int64_t m_sb, int64_t m_sd, const float* z, int64_t z_sb, int64_t z_sd, const float* gamma, const float* beta,
                   const float* mu, const float* rstd, float* dm, int64_t dm_sb, int64_t dm_sd, float* dz, int64_t dz_sb,
                   int64_t dz_sd, float* ws, int D, int L, int npb) {
#define MM_LN_BWDC(NW_, CPL_) hipLaunchKernelGGL((ln_gate_bwdc_kernel<16, NW_, CPL_>), grid, dim3(NW_ * 64), 4 * D * sizeof(float), s, MM_LN_ARGS_B)
#define MM_LN_BWDC32(CPL_) hipLaunchKernelGGL((ln_gate_bwdc_kernel<32, 8, CPL_>), grid, dim3(512), 4 * D * sizeof(float), s, MM_LN_ARGS_B)
  if (pl.pw == 64) {
#define MM_LN_BWDC64(CPL_) hipLaunchKernelGGL((ln_gate_bwdc_kernel<64, 4, CPL_>), grid, dim3(256), 4 * D * sizeof(float), s, MM_LN_ARGS_B)
    if (pl.cpl == 16) MM_LN_BWDC64(16); else if (pl.cpl == 24) MM_LN_BWDC64(24); else MM_LN_BWDC64(32);
#undef MM_LN_BWDC64
  } else if (pl.pw == 32) {
    if (pl.cpl == 12) MM_LN_BWDC32(12); else if (pl.cpl == 16) MM_LN_BWDC32(16); else MM_LN_BWDC32(24);
  } else if (pl.nw == 4) {
    if (pl.cpl == 16) MM_LN_BWDC(4, 16); else MM_LN_BWDC(4, 24);
  } else {
    if (pl.cpl == 16) MM_LN_BWDC(8, 16); else MM_LN_BWDC(8, 24);
  }
#undef MM_LN_BWDC
  return (int)hipGetLastError();
}
#undef MM_LN_ARGS_B

template <int PW>
int launch_ln_fwd1(int cpl, dim3 grid, hipStream_t s, const float* m, int64_t m_sb, int64_t m_sd, const float* z, int64_t z_sb,
                   int64_t z_sd, const float* gamma, const float* beta, float eps, float* y, int64_t y_sb, int64_t y_sd, float* mu,
                   float* rstd, int D, int L, int npb) {
#define MM_LN_FWD1(CPL_) hipLaunchKernelGGL((ln_gate_fwd1_kernel<PW, CPL_>), grid, dim3(256), 0, s, m, m_sb, m_sd, z, z_sb, z_sd, \
                                            gamma, beta, eps, y, y_sb, y_sd, mu, rstd, D, L, npb)
  switch (cpl) {
    case 8: MM_LN_FWD1(8); break;
    case 16: MM_LN_FWD1(16); break;
    case 24: MM_LN_FWD1(24); break;
    case 32: MM_LN_FWD1(32); break;
    default: MM_LN_FWD1(48); break;
  }
#undef MM_LN_FWD1
  return (int)hipGetLastError();
}

template <int PW>
int launch_ln_bwd1(int cpl, dim3 grid, hipStream_t s, const float* dy, int64_t dy_sb, int64_t dy_sd, const float* m, int64_t m_sb,
                   int64_t m_sd, const float* z, int64_t z_sb, int64_t z_sd, const float* gamma, const float* beta, const float* mu,
                   const float* rstd, float* dm, int64_t dm_sb, int64_t dm_sd, float* dz, int64_t dz_sb, int64_t dz_sd, float* ws,
                   int D, int L, int npb) {
#define MM_LN_BWD1(CPL_) hipLaunchKernelGGL((ln_gate_bwd1_kernel<PW, CPL_>), grid, dim3(256), 8 * D * sizeof(float), s, dy, dy_sb, dy_sd, m, m_sb, m_sd, \
                                            z, z_sb, z_sd, gamma, beta, mu, rstd, dm, dm_sb, dm_sd, dz, dz_sb, dz_sd, ws, D, L, npb)
  switch (cpl) {
    case 8: MM_LN_BWD1(8); break;
    case 16: MM_LN_BWD1(16); break;
    default: MM_LN_BWD1(24); break;
  }
#undef MM_LN_BWD1
  return (int)hipGetLastError();
}
}  // namespace

extern "C" {

int mm_dwconv_silu_cross_supported(int H, int W) {   // row strips: any plane a strip of which fits the LDS (W up to ~8000)
  if (H <= 0 || W <= 0) return 0;
  return dw_plan(H, W).lds_bwd <= 150 * 1024 ? 1 : 0;
}
int mm_dwconv_silu_cross_strips(int H, int W) { return (H > 0 && W > 0) ? dw_plan(H, W).nstrips : 0; }

int mm_dwconv_silu_cross_fwd(const float* x, int64_t x_sb, int64_t x_sd, const float* w, const float* bias, float* u2,
                             int64_t u2_sb, int64_t u2_sd, int batch, int D, int H, int W, void* stream) {
  if (!x || !w || !u2) return MM_ERR_NULL;
  if (batch <= 0 || D <= 0 || H <= 0 || W <= 0) return MM_ERR_SHAPE;
  const DwPlan pl = dw_plan(H, W);
  if (pl.lds_bwd > 150 * 1024) return MM_ERR_UNSUPPORTED;
  if ((int64_t)H * W <= 256) {
    const size_t lds = sizeof(float) * 4 * ((size_t)(H + 2) * (W + 2) + (size_t)H * (W + 1));
    hipLaunchKernelGGL(dwconv_silu_cross_fwd_small_kernel, dim3(dw_small_grid(batch * D)), dim3(256), lds, (hipStream_t)stream, x, x_sb, x_sd, w, bias,
                       u2, u2_sb, u2_sd, D, H, W, batch * D);
    return (int)hipGetLastError();
  }
  const int L = pl.SH * W, nt = L >= 1024 ? 256 : (L > 64 ? 128 : 64);
  if (pl.lds_fwd > 64 * 1024) (void)hipFuncSetAttribute((const void*)dwconv_silu_cross_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_fwd);
  hipLaunchKernelGGL(dwconv_silu_cross_fwd_kernel, dim3(batch * D * pl.nstrips), dim3(nt), pl.lds_fwd, (hipStream_t)stream, x, x_sb, x_sd, w, bias, u2,
                     u2_sb, u2_sd, D, H, W, pl.SH, pl.nstrips);
  return (int)hipGetLastError();
}

int mm_dwconv_silu_cross_bwd(const float* du2, int64_t du2_sb, int64_t du2_sd, const float* du4, int64_t du4_sb, int64_t du4_sd,
                             const float* x, int64_t x_sb, int64_t x_sd,
                             const float* w, const float* bias, float* dx, int64_t dx_sb, int64_t dx_sd, float* ws, int batch,
                             int D, int H, int W, void* stream) {
  if (!du2 || !x || !w || !dx || !ws) return MM_ERR_NULL;
  if (batch <= 0 || D <= 0 || H <= 0 || W <= 0) return MM_ERR_SHAPE;
  const DwPlan pl = dw_plan(H, W);
  if (pl.lds_bwd > 150 * 1024) return MM_ERR_UNSUPPORTED;
  if ((int64_t)H * W <= 256) {
    const size_t lds = sizeof(float) * 4 * (2 * (size_t)(H + 2) * (W + 2) + (size_t)H * (W + 1));
    hipLaunchKernelGGL(dwconv_silu_cross_bwd_small_kernel, dim3(dw_small_grid(batch * D)), dim3(256), lds, (hipStream_t)stream, du2, du2_sb, du2_sd,
                       du4, du4_sb, du4_sd, x, x_sb, x_sd, w, bias, dx, dx_sb, dx_sd, ws, D, H, W, batch * D);
    return (int)hipGetLastError();
  }
  const int L = pl.SH * W, nt = L >= 1024 ? 256 : (L > 64 ? 128 : 64);
  if (pl.lds_bwd > 60 * 1024) (void)hipFuncSetAttribute((const void*)dwconv_silu_cross_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bwd);
  hipLaunchKernelGGL(dwconv_silu_cross_bwd_kernel, dim3(batch * D * pl.nstrips), dim3(nt), pl.lds_bwd, (hipStream_t)stream, du2, du2_sb, du2_sd, du4,
                     du4_sb, du4_sd, x, x_sb, x_sd, w, bias, dx, dx_sb, dx_sd, ws, D, H, W, pl.SH, pl.nstrips);
  return (int)hipGetLastError();
}

int mm_cross_merge_fwd(const float* out4, float* m, int64_t m_sb, int64_t m_sd, int batch, int D, int H, int W, void* stream) {
  if (!out4 || !m) return MM_ERR_NULL;
  if (batch <= 0 || D <= 0 || H <= 0 || W <= 0) return MM_ERR_SHAPE;
  if ((int64_t)H * W <= 256)
    hipLaunchKernelGGL(plane_small_kernel<true>, dim3(small_plane_grid(batch * D)), dim3(256), 0, (hipStream_t)stream, out4, (int64_t)0,
                       (int64_t)0, m, m_sb, m_sd, D, H, W, batch * D);
  else
    hipLaunchKernelGGL(cross_merge_fwd_kernel, dim3(((W + 31) / 32) * ((H + 31) / 32) * batch * D), dim3(256), 0, (hipStream_t)stream,
                       out4, m, m_sb, m_sd, D, H, W);
  return (int)hipGetLastError();
}

int mm_plane_transpose(const float* src, int64_t src_sb, int64_t src_sd, float* dst, int64_t dst_sb, int64_t dst_sd, int batch,
                       int D, int H, int W, void* stream) {
  if (!src || !dst) return MM_ERR_NULL;
  if (batch <= 0 || D <= 0 || H <= 0 || W <= 0) return MM_ERR_SHAPE;
  if ((int64_t)H * W <= 256)
    hipLaunchKernelGGL(plane_small_kernel<false>, dim3(small_plane_grid(batch * D)), dim3(256), 0, (hipStream_t)stream, src, src_sb,
                       src_sd, dst, dst_sb, dst_sd, D, H, W, batch * D);
  else
    hipLaunchKernelGGL(plane_transpose_kernel, dim3(((W + 31) / 32) * ((H + 31) / 32) * batch * D), dim3(256), 0, (hipStream_t)stream,
                       src, src_sb, src_sd, dst, dst_sb, dst_sd, D, H, W);
  return (int)hipGetLastError();
}

int mm_ln_gate_rows(int batch, int D, int L) {   // rows of the dgamma/dbeta workspace written by mm_ln_gate_bwd
  const int ppb = plan_ln(batch, D, L, true).ppb;
  return batch * ((L + ppb - 1) / ppb);
}

int mm_ln_gate_fwd(const float* m, int64_t m_sb, int64_t m_sd, const float* z, int64_t z_sb, int64_t z_sd, const float* gamma,
                   const float* beta, float eps, float* y, int64_t y_sb, int64_t y_sd, float* mu, float* rstd, int batch, int D,
                   int L, void* stream) {
  if (!m || !z || !gamma || !beta || !y || !mu || !rstd) return MM_ERR_NULL;
  if (batch <= 0 || D <= 0 || L <= 0) return MM_ERR_SHAPE;
  const LnPlan pl = plan_ln(batch, D, L, false);
  const int pw = pl.pw, npb = (L + pl.ppb - 1) / pl.ppb;
  const dim3 grid(batch * npb), blk(256);
  hipStream_t s = (hipStream_t)stream;
  // 32-bit byte offsets inside one step of the register-resident kernels (rows_rsrc): up to 16 channel rows and one position row
  if ((pl.nw > 0 || pl.cpl > 0) && !ln_offsets_fit(batch, L, {m_sd, z_sd, y_sd})) return MM_ERR_SHAPE;
  if (pl.nw > 0) {
    // channel-major planes (batch stride = L everywhere): the positions of ALL batch items are one contiguous, 128-B aligned row of
    // batch*L floats per channel.  Tiled per item, a 16-position run starts at byte 4*(b*L + 16*k) — at L = 196 / 49 three out of
    // four straddle a 64-B sector, and every item ends in a ragged tile.  Tile the flattened row instead (same grid: the
    // workgroups beyond ceil(batch*L / positions per workgroup) find no position and write nothing).
    if (batch > 1 && m_sb == L && z_sb == L && y_sb == L && ln_flat())
      return launch_ln_fwdc(pl, grid, s, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, eps, y, y_sb, y_sd, mu, rstd, D, batch * L, (int)grid.x);
    return launch_ln_fwdc(pl, grid, s, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, eps, y, y_sb, y_sd, mu, rstd, D, L, npb);
  }
  if (pl.cpl > 0) {
    return pw == 16 ? launch_ln_fwd1<16>(pl.cpl, grid, s, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, eps, y, y_sb, y_sd, mu, rstd, D, L, npb)
                    : launch_ln_fwd1<4>(pl.cpl, grid, s, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, eps, y, y_sb, y_sd, mu, rstd, D, L, npb);
  }
  if (pw == 64) hipLaunchKernelGGL(ln_gate_fwd_kernel<64>, grid, blk, 0, s, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, eps, y, y_sb, y_sd, mu, rstd, D, L, npb);
  else if (pw == 16) hipLaunchKernelGGL(ln_gate_fwd_kernel<16>, grid, blk, 0, s, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, eps, y, y_sb, y_sd, mu, rstd, D, L, npb);
  else hipLaunchKernelGGL(ln_gate_fwd_kernel<4>, grid, blk, 0, s, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, eps, y, y_sb, y_sd, mu, rstd, D, L, npb);
  return (int)hipGetLastError();
}

int mm_ln_gate_bwd(const float* dy, int64_t dy_sb, int64_t dy_sd, const float* m, int64_t m_sb, int64_t m_sd, const float* z,
                   int64_t z_sb, int64_t z_sd, const float* gamma, const float* beta, const float* mu, const float* rstd,
                   float* dm, int64_t dm_sb, int64_t dm_sd, float* dz, int64_t dz_sb, int64_t dz_sd, float* ws, int batch,
                   int D, int L, void* stream) {
  if (!dy || !m || !z || !gamma || !beta || !mu || !rstd || !dm || !dz || !ws) return MM_ERR_NULL;
  if (batch <= 0 || D <= 0 || L <= 0) return MM_ERR_SHAPE;
  const LnPlan pl = plan_ln(batch, D, L, true);
  const int pw = pl.pw, npb = (L + pl.ppb - 1) / pl.ppb;
  const dim3 grid(batch * npb), blk(256);
  hipStream_t s = (hipStream_t)stream;
  if (pl.nw > 0 && !ln_offsets_fit(batch, L, {dy_sd, m_sd, z_sd, dm_sd, dz_sd})) return MM_ERR_SHAPE;
  if (pl.nw > 0) {
    // flattened position row for channel-major planes (see mm_ln_gate_fwd); the grid — and with it the number of partial rows in
    // ws, mm_ln_gate_rows — is unchanged: surplus workgroups contribute zero rows
    if (batch > 1 && dy_sb == L && m_sb == L && z_sb == L && dm_sb == L && dz_sb == L && ln_flat())
      return launch_ln_bwdc(pl, grid, s, dy, dy_sb, dy_sd, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, mu, rstd, dm, dm_sb, dm_sd, dz, dz_sb, dz_sd, ws, D, batch * L, (int)grid.x);
    return launch_ln_bwdc(pl, grid, s, dy, dy_sb, dy_sd, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, mu, rstd, dm, dm_sb, dm_sd, dz, dz_sb, dz_sd, ws, D, L, npb);
  }
  if (pl.cpl > 0) {
    return pw == 16 ? launch_ln_bwd1<16>(pl.cpl, grid, s, dy, dy_sb, dy_sd, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, mu, rstd, dm, dm_sb, dm_sd, dz, dz_sb, dz_sd, ws, D, L, npb)
                    : launch_ln_bwd1<4>(pl.cpl, grid, s, dy, dy_sb, dy_sd, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, mu, rstd, dm, dm_sb, dm_sd, dz, dz_sb, dz_sd, ws, D, L, npb);
  }
  if (pw == 64) hipLaunchKernelGGL(ln_gate_bwd_kernel<64>, grid, blk, 8 * D * sizeof(float), s, dy, dy_sb, dy_sd, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, mu, rstd, dm, dm_sb, dm_sd, dz, dz_sb, dz_sd, ws, D, L, npb);
  else if (pw == 16) hipLaunchKernelGGL(ln_gate_bwd_kernel<16>, grid, blk, 8 * D * sizeof(float), s, dy, dy_sb, dy_sd, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, mu, rstd, dm, dm_sb, dm_sd, dz, dz_sb, dz_sd, ws, D, L, npb);
  else hipLaunchKernelGGL(ln_gate_bwd_kernel<4>, grid, blk, 8 * D * sizeof(float), s, dy, dy_sb, dy_sd, m, m_sb, m_sd, z, z_sb, z_sd, gamma, beta, mu, rstd, dm, dm_sb, dm_sd, dz, dz_sb, dz_sd, ws, D, L, npb);
  return (int)hipGetLastError();
}

}  // extern "C"

// =====================================================================================================
// Block prologue of SS_Conv_SSM.forward (MedMamba.py:350-352): one pass splits the NHWC block input into
//   left  -> NCHW (batch, C2, P) for the conv branch           (replaces chunk + permute(0,3,1,2).contiguous())
//   right -> LayerNorm over its C2 channels, NHWC (batch, P, C2) (replaces chunk + ln_1 on a strided view)
// and the backward writes both halves of d(input) in place (no cat).  ATen's LayerNorm runs at < 0.4 TB/s on
// 48-channel rows; here 16 (or 64) lanes share a row, the row lives in registers, exact two-pass statistics.
// =====================================================================================================
namespace {

template <int TPR>
__device__ __forceinline__ float row_sum(float v) {
  v = group_sum<16>(v);
  if constexpr (TPR == 64) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); }
  return v;
}

constexpr int kLnNV = 8;   // most channels per lane (C2 <= 8 * TPR); the kernels are instantiated for NV = 3, 6, 8 slots per lane: 48 / 96 (16
                           // lanes per row) and 192 / 384 channels (64) of the S / T stages need 3 / 6 — the idle slots still cost their instructions

// (device bodies with a virtual block index / grid size: mm_block_split_* runs the transpose of the left half and the LayerNorm of
//  the right half — independent of each other — as the two block ranges of ONE launch)
template <int TPR, int NV>
__device__ __forceinline__ void ln_half_fwd_body(int vblk, int vgrid, const float* __restrict__ inp, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float eps, float* __restrict__ rn,
                                                 float* __restrict__ mu_out, float* __restrict__ rstd_out,
                                                 int64_t nrows, int C, int C2) {
  constexpr int RPW = 64 / TPR;
  const int lane = threadIdx.x & 63, lr = lane % TPR, lrow = lane / TPR;
  const int64_t wave_global = (int64_t)vblk * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = (int64_t)vgrid * 4;
  const rsrc_t rgam = make_rsrc(gamma, (int64_t)C2 * 4), rbet = make_rsrc(beta, (int64_t)C2 * 4);
  float gm[NV], bt[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    gm[k] = ldb(rgam, (lr + k * TPR) * 4);           // beyond C2: 0
    bt[k] = ldb(rbet, (lr + k * TPR) * 4);
  }
  for (int64_t rg = wave_global; rg * RPW < nrows; rg += nwaves) {
    const int64_t row0 = rg * RPW, row = row0 + lrow;               // row0: wave-uniform
    const bool rok = row < nrows;
    // the RPW rows of this wave through one descriptor (no branch around a load: all NV loads of a lane in flight, §4.4)
    const int nr = (int)(nrows - row0 < RPW ? nrows - row0 : RPW);
    const rsrc_t rx = make_rsrc(inp + row0 * C + C2, ((int64_t)(nr - 1) * C + C2) * 4);
    float x[NV], s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = lr + k * TPR;
      x[k] = ldb(rx, (rok && c < C2) ? (lrow * C + c) * 4 : kOOB);
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) s += x[k];
    const float mean = row_sum<TPR>(s) / C2;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const float dlt = (lr + k * TPR < C2) ? x[k] - mean : 0.f;
      v = fmaf(dlt, dlt, v);
    }
    const float rstd = __builtin_amdgcn_rsqf(row_sum<TPR>(v) / C2 + eps);
    if (rok) {
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = lr + k * TPR;
        if (c < C2) rn[row * C2 + c] = (x[k] - mean) * rstd * gm[k] + bt[k];
      }
      if (lr == 0) { mu_out[row] = mean; rstd_out[row] = rstd; }
    }
  }
}

// d_inp[row, C2 + c] = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = d_rn * gamma;  per-wave partial dgamma / dbeta rows.
template <int TPR, int NV>
__device__ __forceinline__ void ln_half_bwd_body(int vblk, int vgrid, const float* __restrict__ drn, const float* __restrict__ inp,
                                                 const float* __restrict__ gamma, const float* __restrict__ mu_in,
                                                 const float* __restrict__ rstd_in, const float* __restrict__ dres,
                                                 float* __restrict__ dinp, float* __restrict__ ws, int64_t nrows, int C,
                                                 int C2) {
  constexpr int RPW = 64 / TPR;
  const int lane = threadIdx.x & 63, lr = lane % TPR, lrow = lane / TPR;
  const int64_t wave_global = (int64_t)vblk * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = (int64_t)vgrid * 4;
  const rsrc_t rgam = make_rsrc(gamma, (int64_t)C2 * 4);
  float gm[NV], ag[NV], ab[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    gm[k] = ldb(rgam, (lr + k * TPR) * 4);           // beyond C2: 0
    ag[k] = 0.f; ab[k] = 0.f;
  }
  for (int64_t rg = wave_global; rg * RPW < nrows; rg += nwaves) {
    const int64_t row0 = rg * RPW, row = row0 + lrow;               // row0: wave-uniform
    const bool rok = row < nrows;
    // the RPW rows of this wave through one descriptor per operand: 2-3 x NV loads of a lane in flight together (§4.4)
    const int nr = (int)(nrows - row0 < RPW ? nrows - row0 : RPW);
    const int64_t half = ((int64_t)(nr - 1) * C + C2) * 4;
    const rsrc_t rd = make_rsrc(drn + row0 * C2, (int64_t)nr * C2 * 4), rx = make_rsrc(inp + row0 * C + C2, half),
                 rr = make_rsrc(dres ? dres + row0 * C + C2 : nullptr, dres ? half : 0), rst = make_rsrc(mu_in + row0, (int64_t)nr * 4),
                 rrs = make_rsrc(rstd_in + row0, (int64_t)nr * 4);
    const float mean = ldb(rst, lrow * 4), rstd = ldb(rrs, lrow * 4);       // rows beyond nrows: 0
    float xh[NV], g[NV], dv[NV], rv[NV], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int c = lr + k * TPR;
      const bool ok = rok && c < C2;
      dv[k] = ldb(rd, ok ? (lrow * C2 + c) * 4 : kOOB);
      xh[k] = ldb(rx, ok ? (lrow * C + c) * 4 : kOOB);
      rv[k] = ldb(rr, ok ? (lrow * C + c) * 4 : kOOB);
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const bool ok = rok && lr + k * TPR < C2;
      const float d = dv[k];
      xh[k] = ok ? (xh[k] - mean) * rstd : 0.f;
      g[k] = d * gm[k];
      s1 += g[k];
      s2 = fmaf(g[k], xh[k], s2);
      ag[k] = fmaf(d, xh[k], ag[k]);
      ab[k] += d;
    }
    const float c1 = row_sum<TPR>(s1) / C2, c2 = row_sum<TPR>(s2) / C2;
    if (rok) {
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int c = lr + k * TPR;
        if (c < C2) dinp[row * C + C2 + c] = rstd * (g[k] - c1 - xh[k] * c2) + rv[k];
      }
    }
  }
  // fold the RPW row slots of the wave (lanes with equal lr), then the 4 waves through LDS: one partial row per workgroup
  extern __shared__ float sred[];                       // [4 waves][2*C2]: one row per wave (plain stores), summed in wave order
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float a = ag[k], bsum = ab[k];
    if constexpr (TPR == 16) {
      a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
      bsum += __shfl_xor(bsum, 16); bsum += __shfl_xor(bsum, 32);
    }
    const int c = lr + k * TPR;
    if (lane < TPR && c < C2) { sred[wv * 2 * C2 + c] = a; sred[wv * 2 * C2 + C2 + c] = bsum; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C2; i += blockDim.x)
    ws[(int64_t)vblk * 2 * C2 + i] = (sred[i] + sred[2 * C2 + i]) + (sred[4 * C2 + i] + sred[6 * C2 + i]);
}

// dst[b, i, p] = src[b, p, i] for i < C2 (src row stride C): NHWC half -> NCHW.  REV: the other way round
// (dst[b, p, i] = src[b, i, p] written into a buffer with row stride C).  grid: ceil(P/32) * ceil(C2/32) * batch
// REV only: `add` (same layout as dst, or null) is added to what is written (gradient of the residual path).
template <bool REV>
__device__ __forceinline__ void half_transpose_body(int vblk, const float* __restrict__ src, float* __restrict__ dst,
                                                    const float* __restrict__ add, int P, int C, int C2) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int nbp = (P + 31) / 32, nbi = (C2 + 31) / 32;
  const int p0 = (vblk % nbp) * 32, i0 = ((vblk / nbp) % nbi) * 32, b = vblk / (nbp * nbi);
  // tile loads through a descriptor that starts at the tile (32-bit offsets for any tensor size), no branch around a load: the four
  // loads of a thread are in flight together (§4.4)
  if constexpr (!REV) {
    const rsrc_t rs = make_rsrc(src + ((int64_t)b * P + p0) * C + i0, ((int64_t)(min(32, P - p0) - 1) * C + min(32, C2 - i0)) * 4);
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                // rows p, lanes along i
      const int r = ty + 8 * q;
      v[q] = ldb(rs, (p0 + r < P && i0 + tx < C2) ? (r * C + tx) * 4 : kOOB);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) tile[ty + 8 * q][tx] = v[q];
    __syncthreads();
    float* d = dst + (int64_t)b * C2 * P;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {           // rows i, lanes along p
      const int i = i0 + r, p = p0 + tx;
      // `add` doubles as the optional per-channel affine [scale C2 | shift C2] of the forward direction (eval-mode
      // BatchNorm in front of the conv branch folded into this copy: applied BEFORE the conv's zero padding, hence exact)
      if (i < C2 && p < P) d[(int64_t)i * P + p] = add ? fmaf(tile[tx][r], add[i], add[C2 + i]) : tile[tx][r];
    }
  } else {
    const rsrc_t rs = make_rsrc(src + ((int64_t)b * C2 + i0) * P + p0, ((int64_t)(min(32, C2 - i0) - 1) * P + min(32, P - p0)) * 4);
    const rsrc_t ra = make_rsrc(add ? add + ((int64_t)b * P + p0) * C + i0 : nullptr,
                                add ? ((int64_t)(min(32, P - p0) - 1) * C + min(32, C2 - i0)) * 4 : 0);
    float v[4], av[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty + 8 * q;
      v[q] = ldb(rs, (i0 + r < C2 && p0 + tx < P) ? (r * P + tx) * 4 : kOOB);
      av[q] = ldb(ra, (p0 + r < P && i0 + tx < C2) ? (r * C + tx) * 4 : kOOB);      // the element this thread stores below
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) tile[ty + 8 * q][tx] = v[q];
    __syncthreads();
    float* d = dst + (int64_t)b * P * C;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty + 8 * q, p = p0 + r, i = i0 + tx;
      if (p < P && i < C2) d[(int64_t)p * C + i] = tile[tx][r] + av[q];
    }
  }
}

// blocks [0, nT): NHWC left half -> NCHW (optionally with the folded BatchNorm affine); blocks [nT, grid): ln_1 of the right half
template <int TPR, int NV>
__global__ __launch_bounds__(256) void block_split_fwd_kernel(int nT, const float* __restrict__ inp, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              const float* __restrict__ left_affine, float* __restrict__ left_nchw,
                                                              float* __restrict__ rn, float* __restrict__ mu, float* __restrict__ rstd,
                                                              int64_t nrows, int P, int C, int C2) {
  if ((int)blockIdx.x < nT) half_transpose_body<false>(blockIdx.x, inp, left_nchw, left_affine, P, C, C2);
  else ln_half_fwd_body<TPR, NV>(blockIdx.x - nT, gridDim.x - nT, inp, gamma, beta, eps, rn, mu, rstd, nrows, C, C2);
}
// blocks [0, nT): d(left) NCHW -> d(inp)[..., :C2] (+ residual gradient); blocks [nT, grid): LayerNorm backward into d(inp)[..., C2:]
template <int TPR, int NV>
__global__ __launch_bounds__(256) void block_split_bwd_kernel(int nT, const float* __restrict__ dleft_nchw, const float* __restrict__ drn,
                                                              const float* __restrict__ dres, const float* __restrict__ inp,
                                                              const float* __restrict__ gamma, const float* __restrict__ mu,
                                                              const float* __restrict__ rstd, float* __restrict__ dinp,
                                                              float* __restrict__ ws, int64_t nrows, int P, int C, int C2) {
  if ((int)blockIdx.x < nT) half_transpose_body<true>(blockIdx.x, dleft_nchw, dinp, dres, P, C, C2);
  else ln_half_bwd_body<TPR, NV>(blockIdx.x - nT, gridDim.x - nT, drn, inp, gamma, mu, rstd, dres, dinp, ws, nrows, C, C2);
}

inline int ln_half_grid(int64_t nrows, int tpr) {
  const int64_t waves = (nrows + (64 / tpr) - 1) / (64 / tpr);
  int64_t blocks = (waves + 3) / 4;
  if (blocks > 1024) blocks = 1024;       // >= 16 waves per CU; every workgroup leaves ONE partial dgamma/dbeta row
  return (int)blocks;
}
}  // namespace

extern "C" {

int mm_block_split_rows(int batch, int P, int C2) {   // rows of the dgamma/dbeta workspace of mm_block_split_bwd
  return ln_half_grid((int64_t)batch * P, C2 <= 128 ? 16 : 64);
}

int mm_block_split_fwd(const float* inp, const float* gamma, const float* beta, float eps, const float* left_affine,
                       float* left_nchw, float* rn, float* mu, float* rstd, int batch, int P, int C2, void* stream) {
  if (!inp || !gamma || !beta || !left_nchw || !rn || !mu || !rstd) return MM_ERR_NULL;
  if (batch <= 0 || P <= 0 || C2 <= 0) return MM_ERR_SHAPE;
  if (C2 > 8 * 64) return MM_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int C = 2 * C2;
  const int64_t nrows = (int64_t)batch * P;
  const int nT = ((P + 31) / 32) * ((C2 + 31) / 32) * batch;
  const int tpr = C2 <= 128 ? 16 : 64, need = (C2 + tpr - 1) / tpr;
  const dim3 grid(nT + ln_half_grid(nrows, tpr));
#define MM_BS_FWD(TPR_, NV_) hipLaunchKernelGGL((block_split_fwd_kernel<TPR_, NV_>), grid, dim3(256), 0, s, nT, inp, gamma, beta, eps, left_affine, left_nchw, rn, mu, rstd, nrows, P, C, C2)
  if (tpr == 16) { if (need <= 3) MM_BS_FWD(16, 3); else if (need <= 6) MM_BS_FWD(16, 6); else MM_BS_FWD(16, 8); }
  else { if (need <= 3) MM_BS_FWD(64, 3); else if (need <= 6) MM_BS_FWD(64, 6); else MM_BS_FWD(64, 8); }
#undef MM_BS_FWD
  return (int)hipGetLastError();
}

int mm_block_split_bwd(const float* dleft_nchw, const float* drn, const float* dres, const float* inp, const float* gamma,
                       const float* mu, const float* rstd, float* dinp, float* ws, int batch, int P, int C2, void* stream) {
  if (!dleft_nchw || !drn || !inp || !gamma || !mu || !rstd || !dinp || !ws) return MM_ERR_NULL;
  if (batch <= 0 || P <= 0 || C2 <= 0) return MM_ERR_SHAPE;
  if (C2 > 8 * 64) return MM_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int C = 2 * C2;
  const int64_t nrows = (int64_t)batch * P;
  const int nT = ((P + 31) / 32) * ((C2 + 31) / 32) * batch;
  const int tpr = C2 <= 128 ? 16 : 64, need = (C2 + tpr - 1) / tpr;
  const dim3 grid(nT + ln_half_grid(nrows, tpr));
#define MM_BS_BWD(TPR_, NV_) hipLaunchKernelGGL((block_split_bwd_kernel<TPR_, NV_>), grid, dim3(256), 8 * C2 * sizeof(float), s, nT, dleft_nchw, drn, dres, inp, gamma, mu, rstd, dinp, ws, nrows, P, C, C2)
  if (tpr == 16) { if (need <= 3) MM_BS_BWD(16, 3); else if (need <= 6) MM_BS_BWD(16, 6); else MM_BS_BWD(16, 8); }
  else { if (need <= 3) MM_BS_BWD(64, 3); else if (need <= 6) MM_BS_BWD(64, 6); else MM_BS_BWD(64, 8); }
#undef MM_BS_BWD
  return (int)hipGetLastError();
}

}  // extern "C"

// =====================================================================================================
// SS2D parameter packing: the five direction-indexed parameters (reference order k = row fwd, col fwd, row rev,
// col rev: MedMamba.py:256-257) -> one buffer in KERNEL direction order g = (row fwd, row rev, col fwd, col rev),
// with A = -exp(A_logs) (MedMamba.py:271).  k(g) = (0,2,1,3) is an involution, so the gradient un-packing is the
// same index map.  Layout of the packed buffer (floats): [Wx 4*C*D | Wdt 4*D*R | A 4*D*N | D 4*D | bias 4*D], every
// segment padded to a multiple of 64 floats (the GEMM libraries pick slower kernels for operands off 256-B alignment).
// =====================================================================================================
namespace {
struct PackSeg { int per_dir[5]; int off[6]; };

__host__ __device__ __forceinline__ PackSeg pack_layout(int D, int C, int R, int N) {
  PackSeg s;
  s.per_dir[0] = C * D; s.per_dir[1] = D * R; s.per_dir[2] = D * N; s.per_dir[3] = D; s.per_dir[4] = D;
  s.off[0] = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) s.off[i + 1] = (s.off[i] + 4 * s.per_dir[i] + 63) & ~63;   // 256-B aligned segments
  return s;
}

// BWD = false: P[packed] = f(src_seg[k-order]);  BWD = true: G[k-order, same segment layout] = dP[packed] * (A-seg ? P : 1)
// One workgroup = 256 consecutive elements of one (segment, direction): the index math is wave-uniform (no per-thread
// division), reads and writes are contiguous runs.  blk_off[s] = first workgroup of segment s, nb[s] = workgroups per
// direction of segment s (computed on the host).
struct PackGrid { int blk_off[6]; int nb[5]; };

// Small reductions of the SS2D backward that ride on the un-packing launch (mm_ss2d_pack_bwd): the per-workgroup partial rows of
// ln_gate_bwd (dgamma | dbeta of out_norm) and the per-(plane, strip) partial sums of the depthwise conv's weight / bias
// gradient, written in their final contiguous layouts behind the packed gradients — 2 reduction launches and 2 layout copies per
// block less on the main stream.
struct PackExtra {
  const float* ln_ws;      // (ln_rows, 2*D) or nullptr
  const float* dw_ws;      // (batch, D*S, 10) or nullptr
  float* ln_out;           // 2*D: dgamma | dbeta
  float* dw_out;           // D*9 (weight gradient, (D,1,3,3) order) | D (bias gradient)
  int ln_rows, ln_rl, dw_batch, dw_S, blk_ln, blk_dw;    // blocks [blk_ln, blk_dw) reduce ln_ws, [blk_dw, grid) reduce dw_ws
};

__device__ __forceinline__ void pack_extra_blocks(const PackExtra& ex, int D, float* red) {
  const int b = blockIdx.x;
  if (b < ex.blk_dw) {
    // ln: block = (256 / RL) columns x RL row lanes (RL = ex.ln_rl: 4 for a few hundred rows, 32 for the 3136 rows of the long-
    // sequence stages, where 4 lanes walked 784 rows each: 90 us); rows strided by RL, then RL -> 1 through LDS in lane order
    const int RL = ex.ln_rl, COLS = 256 / RL;
    const int col = (b - ex.blk_ln) * COLS + (int)(threadIdx.x % COLS), rl = threadIdx.x / COLS;
    float a0 = 0.f, a1 = 0.f;
    if (col < 2 * D) {
      const float* src = ex.ln_ws + col;
      int r = rl;
      // eight loads in flight, added in the order of the two-row loop below (same bits): with two in flight the 112-784 rows of a
      // lane were 56-392 dependent round trips to L2
      for (; r + 7 * RL < ex.ln_rows; r += 8 * RL) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(r + u * RL) * 2 * D];
        a0 += v[0]; a1 += v[1]; a0 += v[2]; a1 += v[3]; a0 += v[4]; a1 += v[5]; a0 += v[6]; a1 += v[7];
      }
      for (; r + RL < ex.ln_rows; r += 2 * RL) { a0 += src[(int64_t)r * 2 * D]; a1 += src[(int64_t)(r + RL) * 2 * D]; }
      if (r < ex.ln_rows) a0 += src[(int64_t)r * 2 * D];
    }
    red[threadIdx.x] = a0 + a1;
    __syncthreads();
    if (rl == 0 && col < 2 * D) {
      float t = red[threadIdx.x];
      for (int k = 1; k < RL; ++k) t += red[threadIdx.x + k * COLS];
      ex.ln_out[col] = t;
    }
  } else {
    // depthwise conv: 4 lanes per output (channel d, j of 10), each a quarter of the batch with 4 loads in flight, joined by DPP
    // in lane order; sums over batch and strips in a fixed order
    const int o = (b - ex.blk_dw) * 64 + (int)(threadIdx.x >> 2), part = threadIdx.x & 3;
    float acc = 0.f;
    int d = 0, j = 0;
    if (o < D * 10) {
      d = o / 10; j = o - d * 10;
      const int64_t bstride = (int64_t)D * ex.dw_S * 10;
      const int q0 = (ex.dw_batch * part) >> 2, q1 = (ex.dw_batch * (part + 1)) >> 2;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int s = 0; s < ex.dw_S; ++s) {
        const float* src = ex.dw_ws + ((int64_t)d * ex.dw_S + s) * 10 + j;
        int q = q0;
        for (; q + 7 < q1; q += 8) {                   // eight loads in flight, added in the order of the four-wide loop below
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = src[(q + u) * bstride];
          a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3]; a0 += v[4]; a1 += v[5]; a2 += v[6]; a3 += v[7];
        }
        for (; q + 3 < q1; q += 4) {
          a0 += src[q * bstride]; a1 += src[(q + 1) * bstride]; a2 += src[(q + 2) * bstride]; a3 += src[(q + 3) * bstride];
        }
        for (; q < q1; ++q) a0 += src[q * bstride];
      }
      acc = (a0 + a1) + (a2 + a3);
    }
    acc += mm::dpp_f<mm::DPP_QUAD_XOR1>(acc);
    acc += mm::dpp_f<mm::DPP_QUAD_XOR2>(acc);
    if (part == 0 && o < D * 10) {
      if (j < 9) ex.dw_out[d * 9 + j] = acc; else ex.dw_out[D * 9 + d] = acc;
    }
  }
}

template <bool BWD>
__global__ __launch_bounds__(256) void ss2d_pack_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                                                        const float* __restrict__ s2, const float* __restrict__ s3,
                                                        const float* __restrict__ s4, float* __restrict__ dst, int D, int C,
                                                        int R, int N, PackGrid pg, int nparts, PackExtra ex) {
  if constexpr (BWD) {
    __shared__ float red[256];
    if ((int)blockIdx.x >= ex.blk_ln) { pack_extra_blocks(ex, D, red); return; }
  }
  // no runtime-indexed local arrays here: they would live in scratch memory, and a dispatch that needs scratch costs
  // ~12 us of set-up on top of a 2 us kernel (measured)
  const int b = blockIdx.x;
  const int seg = (b >= pg.blk_off[1]) + (b >= pg.blk_off[2]) + (b >= pg.blk_off[3]) + (b >= pg.blk_off[4]);
  const int boff = seg == 0 ? pg.blk_off[0] : seg == 1 ? pg.blk_off[1] : seg == 2 ? pg.blk_off[2] : seg == 3 ? pg.blk_off[3] : pg.blk_off[4];
  const int nbs = seg == 0 ? pg.nb[0] : seg == 1 ? pg.nb[1] : seg == 2 ? pg.nb[2] : seg == 3 ? pg.nb[3] : pg.nb[4];
  const int pd = seg == 0 ? C * D : seg == 1 ? D * R : seg == 2 ? D * N : D;
  const int al = 63;
  const int o1 = (4 * C * D + al) & ~al, o2 = (o1 + 4 * D * R + al) & ~al, o3 = (o2 + 4 * D * N + al) & ~al,
            o4 = (o3 + 4 * D + al) & ~al;
  const int soff = seg == 0 ? 0 : seg == 1 ? o1 : seg == 2 ? o2 : seg == 3 ? o3 : o4;      // == pack_layout().off[seg]
  const int lb = b - boff;
  const int g = lb / nbs, chunk = lb - g * nbs;            // wave-uniform
  const int rem = chunk * 256 + threadIdx.x;
  if (rem >= pd) return;
  const int k = ((g & 1) << 1) | (g >> 1);                 // reference direction of kernel direction g (an involution)
  const int i = soff + g * pd + rem;                       // packed index (kernel order)
  const int j = k * pd + rem;                              // index inside the segment, reference direction order
  if constexpr (!BWD) {
    const float* src = seg == 0 ? s0 : seg == 1 ? s1 : seg == 2 ? s3 : seg == 3 ? s4 : s2;   // (Wx, Wdt, bias, A_logs, Ds)
    const float v = src[j];
    dst[i] = seg == 2 ? -expf(v) : v;
  } else {
    // s0 = dP (packed), s1 = P (packed): d(A_logs) = dA * A.  nparts > 0: the A / D / bias segments of the gradient are
    // the sum over the per-batch-item partial buffers s2[part][i - off(A)] that mm_scan_bwd filled (mm_scan_args.dpar_sb):
    // summed here in a fixed order (no atomics anywhere -> reproducible), the same segments of dP are not read
    float v;
    if (nparts > 0 && seg >= 2) {
      const int o5 = (o4 + 4 * D + al) & ~al;
      const int S = o5 - o2;
      const float* src = s2 + (i - o2);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int q = 0;
      for (; q + 16 <= nparts; q += 16) {                // sixteen loads in flight, added in the order of the four-wide loop below
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[(int64_t)(q + u) * S];
#pragma unroll
        for (int u = 0; u < 16; u += 4) { a0 += v[u]; a1 += v[u + 1]; a2 += v[u + 2]; a3 += v[u + 3]; }
      }
      for (; q + 4 <= nparts; q += 4) {
        a0 += src[(int64_t)q * S]; a1 += src[(int64_t)(q + 1) * S]; a2 += src[(int64_t)(q + 2) * S]; a3 += src[(int64_t)(q + 3) * S];
      }
      for (; q < nparts; ++q) a0 += src[(int64_t)q * S];
      v = (a0 + a1) + (a2 + a3);
    } else {
      v = s0[i];
    }
    dst[soff + j] = v * (seg == 2 ? s1[i] : 1.0f);
  }
}

inline PackGrid pack_grid(int D, int C, int R, int N) {
  const PackSeg L = pack_layout(D, C, R, N);
  PackGrid pg;
  pg.blk_off[0] = 0;
  for (int s = 0; s < 5; ++s) {
    pg.nb[s] = (L.per_dir[s] + 255) / 256;
    pg.blk_off[s + 1] = pg.blk_off[s] + 4 * pg.nb[s];
  }
  return pg;
}
}  // namespace

extern "C" {

int mm_ss2d_pack_size(int D, int C, int R, int N) { return pack_layout(D, C, R, N).off[5]; }

int mm_ss2d_pack_fwd(const float* x_proj_w, const float* dt_w, const float* dt_b, const float* A_logs, const float* Ds,
                     float* packed, int D, int C, int R, int N, void* stream) {
  if (!x_proj_w || !dt_w || !dt_b || !A_logs || !Ds || !packed) return MM_ERR_NULL;
  if (D <= 0 || C <= 0 || R <= 0 || N <= 0) return MM_ERR_SHAPE;
  const PackGrid pg = pack_grid(D, C, R, N);
  PackExtra ex{};
  ex.blk_ln = ex.blk_dw = pg.blk_off[5];
  hipLaunchKernelGGL(ss2d_pack_kernel<false>, dim3(pg.blk_off[5]), dim3(256), 0, (hipStream_t)stream, x_proj_w, dt_w, dt_b,
                     A_logs, Ds, packed, D, C, R, N, pg, 0, ex);
  return (int)hipGetLastError();
}

int mm_ss2d_pack_parts_size(int D, int C, int R, int N) {
  const PackSeg L = pack_layout(D, C, R, N);
  return L.off[5] - L.off[2];
}

int mm_ss2d_pack_bwd(const float* dpacked, const float* packed, const float* parts, float* grads, int D, int C, int R, int N,
                     int nparts, const float* ln_ws, int ln_rows, float* ln_out, const float* dw_ws, int dw_batch, int dw_strips,
                     float* dw_out, void* stream) {
  if (!dpacked || !packed || !grads || (nparts > 0 && !parts)) return MM_ERR_NULL;
  if (D <= 0 || C <= 0 || R <= 0 || N <= 0 || nparts < 0) return MM_ERR_SHAPE;
  if ((ln_ws && (!ln_out || ln_rows <= 0)) || (dw_ws && (!dw_out || dw_batch <= 0 || dw_strips <= 0))) return MM_ERR_SHAPE;
  const PackGrid pg = pack_grid(D, C, R, N);
  PackExtra ex{};
  ex.ln_ws = ln_ws; ex.ln_rows = ln_rows; ex.ln_out = ln_out;
  ex.dw_ws = dw_ws; ex.dw_batch = dw_batch; ex.dw_S = dw_strips; ex.dw_out = dw_out;
  ex.blk_ln = pg.blk_off[5];
  ex.ln_rl = ln_rows > 1024 ? 32 : (ln_rows > 256 ? 16 : 4);
  ex.blk_dw = ex.blk_ln + (ln_ws ? (2 * D + 256 / ex.ln_rl - 1) / (256 / ex.ln_rl) : 0);
  const int grid = ex.blk_dw + (dw_ws ? (D * 10 + 63) / 64 : 0);
  hipLaunchKernelGGL(ss2d_pack_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dpacked, packed, parts,
                     nullptr, nullptr, grads, D, C, R, N, pg, nparts, ex);
  return (int)hipGetLastError();
}

}  // extern "C"

// =====================================================================================================
// Per-channel sum of an NCHW tensor: out[c] = sum_{b,p} x[b,c,p]  — the bias gradient of the conv branch's
// convolutions (MedMamba.py:338-346).  ATen's generic reduction runs this at ~0.65 TB/s (59 us for 64x48x56x56);
// here one workgroup streams one channel (float4 loads, LDS tree), a second grid dimension splits the batch when
// there are fewer channels than CUs (partials then go through fp32 atomics into a zero-filled out).
// =====================================================================================================
namespace {
template <bool VEC, bool ATOMIC>
__global__ __launch_bounds__(512) void channel_sum_nchw_kernel(const float* __restrict__ x, float* __restrict__ out, int batch,
                                                               int C, int HW, int bsplit) {
  __shared__ float red[8];
  const int c = blockIdx.x, part = blockIdx.y;
  const int b0 = (int)((int64_t)batch * part / bsplit), b1 = (int)((int64_t)batch * (part + 1) / bsplit);
  float acc = 0.f;
  for (int b = b0; b < b1; ++b) {
    const float* p = x + ((int64_t)b * C + c) * HW;
    if constexpr (VEC) {
      const float4* p4 = reinterpret_cast<const float4*>(p);
      for (int i = threadIdx.x; i < HW / 4; i += blockDim.x) {
        const float4 v = p4[i];
        acc += (v.x + v.y) + (v.z + v.w);
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += blockDim.x) acc += p[i];
    }
  }
  acc = mm::group_sum<16>(acc);
  acc += __shfl_xor(acc, 16);
  acc += __shfl_xor(acc, 32);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) red[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int j = 0; j < (int)(blockDim.x >> 6); ++j) t += red[j];
    if constexpr (ATOMIC) out[(int64_t)part * C + c] = t; else out[c] = t;     // (ATOMIC = several batch parts: one row per part, no atomics)
  }
}
}  // namespace

extern "C" {

int mm_channel_sum_nchw_split(int batch, int C) {      // > 1: `out` holds that many rows of C partial sums (the caller adds them)
  int s = 1;
  while (C * s < 256 && s * 2 <= batch && s < 16) s *= 2;
  return s;
}

int mm_channel_sum_nchw(const float* x, float* out, int batch, int C, int HW, void* stream) {
  if (!x || !out) return MM_ERR_NULL;
  if (batch <= 0 || C <= 0 || HW <= 0) return MM_ERR_SHAPE;
  const int split = mm_channel_sum_nchw_split(batch, C);
  const bool vec = (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const int nt = HW >= 2048 ? 512 : (HW >= 256 ? 256 : 64);
  const dim3 grid(C, split), blk(nt);
  hipStream_t s = (hipStream_t)stream;
  if (split > 1) {
    if (vec) hipLaunchKernelGGL((channel_sum_nchw_kernel<true, true>), grid, blk, 0, s, x, out, batch, C, HW, split);
    else hipLaunchKernelGGL((channel_sum_nchw_kernel<false, true>), grid, blk, 0, s, x, out, batch, C, HW, split);
  } else {
    if (vec) hipLaunchKernelGGL((channel_sum_nchw_kernel<true, false>), grid, blk, 0, s, x, out, batch, C, HW, split);
    else hipLaunchKernelGGL((channel_sum_nchw_kernel<false, false>), grid, blk, 0, s, x, out, batch, C, HW, split);
  }
  return (int)hipGetLastError();
}

}  // extern "C"
