// Training-mode BatchNorm2d (+ the ReLU that follows it) of the conv branch (MedMamba.py:338-346) on NCHW planes.
//
// Why not MIOpen's: at the 56x56 stage (64 x 48 x 56 x 56, 38.5 MB) MIOpenBatchNormFwdTrainSpatial takes 139 us and the
// backward 74 us (0.55 / 1.5 TB/s of their own traffic), and the ReLU behind two of the three BatchNorms is one more
// elementwise pass each way.  Here: per-(channel, batch-slice) partial statistics in one streaming pass, then one apply
// pass that normalises, scales and clamps — every operand crosses HBM once per pass, ReLU costs nothing.
//
//   forward :  bn_stats_kernel      x -> partial (count, mean, M2) per (slice, channel)      [exact two-pass inside a slice]
//              bn_apply_fwd_kernel  partials merged (Chan) -> mean, rstd, running stats; y = relu?((x-mean)*rstd*gamma+beta)
//   backward:  bn_bwd_stats_kernel  g = dy * [y > 0];  partial (sum g, sum g*xhat) per (slice, channel)
//              bn_apply_bwd_kernel  dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat));  dgamma, dbeta
// Semantics = torch.nn.BatchNorm2d in training mode: biased variance for the normalisation, unbiased for running_var,
// running = (1 - momentum) * running + momentum * batch.  The ReLU mask in the backward is recomputed from x with the forward's
// own expression fmaf(x, rstd*gamma, beta - mean*rstd*gamma) (y is not read).
#include <stdlib.h>
#include "mm_common.h"
#include "medmamba_hip.h"

namespace {
using namespace mm;

__device__ __forceinline__ float block_sum(float v, float* red) {     // sum over a 256-thread workgroup, result in every thread
  v = group_sum<16>(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// slice s of the batch: images [s*nb, min((s+1)*nb, batch))
struct BnGeom { int batch, C, HW, nb, S, SP; };   // SP = number of statistics partials per channel (= S unless they come from elsewhere)

// partial[(s*C + c)*3 + (0: count, 1: mean, 2: M2)]
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ partial, BnGeom g) {
  __shared__ float red[4];
  const int c = blockIdx.x % g.C, s = blockIdx.x / g.C;
  const int b0 = s * g.nb, b1 = min(b0 + g.nb, g.batch);
  const int hw4 = g.HW >> 2;
  float sum = 0.f;
  for (int b = b0; b < b1; ++b) {
    const float* p = x + ((int64_t)b * g.C + c) * g.HW;
    if ((g.HW & 3) == 0) {
      const float4* p4 = reinterpret_cast<const float4*>(p);
      for (int i = threadIdx.x; i < hw4; i += 256) { const float4 v = p4[i]; sum += (v.x + v.y) + (v.z + v.w); }
    } else {
      for (int i = threadIdx.x; i < g.HW; i += 256) sum += p[i];
    }
  }
  const float n = (float)(b1 - b0) * g.HW;
  const float mean = block_sum(sum, red) / n;
  float m2 = 0.f;
  for (int b = b0; b < b1; ++b) {                 // second pass over the slice: it is L2-resident (<= a few hundred KB)
    const float* p = x + ((int64_t)b * g.C + c) * g.HW;
    if ((g.HW & 3) == 0) {
      const float4* p4 = reinterpret_cast<const float4*>(p);
      for (int i = threadIdx.x; i < hw4; i += 256) {
        const float4 v = p4[i];
        const float a = v.x - mean, bb = v.y - mean, cc = v.z - mean, d = v.w - mean;
        m2 += (a * a + bb * bb) + (cc * cc + d * d);
      }
    } else {
      for (int i = threadIdx.x; i < g.HW; i += 256) { const float a = p[i] - mean; m2 = fmaf(a, a, m2); }
    }
  }
  m2 = block_sum(m2, red);
  if (threadIdx.x == 0) {
    float* o = partial + ((int64_t)s * g.C + c) * 3;
    o[0] = n; o[1] = mean; o[2] = m2;
  }
}

// merge the S partials of channel c (Chan et al.): every thread computes the same few values
__device__ __forceinline__ void bn_merge(const float* __restrict__ partial, int c, const BnGeom& g, float& n, float& mean, float& m2) {
  n = 0.f; mean = 0.f; m2 = 0.f;
  for (int s = 0; s < g.SP; ++s) {
    const float* o = partial + ((int64_t)s * g.C + c) * 3;
    const float nb = o[0], mb = o[1], Mb = o[2];
    const float nn = n + nb, d = mb - mean;
    mean += d * (nb / nn);
    m2 += Mb + d * d * (n * nb / nn);
    n = nn;
  }
}

__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ partial,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           float momentum, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ y,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out, int relu,
                                                           const float* __restrict__ pre_bias, BnGeom g) {
  const int c = blockIdx.x % g.C, s = blockIdx.x / g.C;
  float n, mean, m2;
  bn_merge(partial, c, g, n, mean, m2);
  const float var = m2 / n;
  const float rstd = 1.0f / sqrtf(var + eps);
  const float sc = rstd * gamma[c], sh = beta[c] - mean * sc;
  if (s == 0 && threadIdx.x == 0) {
    mean_out[c] = mean; rstd_out[c] = rstd;
    // pre_bias: x is the output of a convolution WITHOUT its bias; BatchNorm(x + bias) == BatchNorm(x) except for the running
    // mean, which follows the biased tensor (the separate bias-add pass over x never runs)
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (mean + (pre_bias ? pre_bias[c] : 0.f));
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1.f ? m2 / (n - 1.f) : var);
  }
  const int b0 = s * g.nb, b1 = min(b0 + g.nb, g.batch);
  const int hw4 = g.HW >> 2;
  for (int b = b0; b < b1; ++b) {
    const int64_t off = ((int64_t)b * g.C + c) * g.HW;
    if ((g.HW & 3) == 0) {
      const float4* p4 = reinterpret_cast<const float4*>(x + off);
      float4* q4 = reinterpret_cast<float4*>(y + off);
      for (int i = threadIdx.x; i < hw4; i += 256) {
        const float4 v = p4[i];
        float4 o = make_float4(fmaf(v.x, sc, sh), fmaf(v.y, sc, sh), fmaf(v.z, sc, sh), fmaf(v.w, sc, sh));
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        q4[i] = o;
      }
    } else {
      for (int i = threadIdx.x; i < g.HW; i += 256) {
        const float o = fmaf(x[off + i], sc, sh);
        y[off + i] = relu ? fmaxf(o, 0.f) : o;
      }
    }
  }
}

// partial2[(s*C + c)*2 + (0: sum g, 1: sum g*xhat)],  g = dy * [relu ? (xhat*gamma+beta > 0) : 1]
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                           float* __restrict__ partial2, BnGeom g) {
  __shared__ float red[4];
  const int c = blockIdx.x % g.C, s = blockIdx.x / g.C;
  const float mean = mean_in[c], rstd = rstd_in[c], gm = gamma[c], bt = beta[c];
  const int b0 = s * g.nb, b1 = min(b0 + g.nb, g.batch);
  const int hw4 = g.HW >> 2;
  // ReLU mask: the SAME expression and rounding as the forward's y = fmaf(x, sc, sh) (bn_apply_fwd_kernel), so an element
  // the forward clamped never passes gradient and vice versa (torch masks on the stored y)
  const float sc = rstd * gm, sh = bt - mean * sc;
  float s1 = 0.f, s2 = 0.f;
  auto acc = [&](float d, float xv) {
    const float xh = (xv - mean) * rstd;
    const float gg = (relu && fmaf(xv, sc, sh) <= 0.f) ? 0.f : d;
    s1 += gg; s2 = fmaf(gg, xh, s2);
  };
  for (int b = b0; b < b1; ++b) {
    const int64_t off = ((int64_t)b * g.C + c) * g.HW;
    if ((g.HW & 3) == 0) {
      const float4* d4 = reinterpret_cast<const float4*>(dy + off);
      const float4* p4 = reinterpret_cast<const float4*>(x + off);
      for (int i = threadIdx.x; i < hw4; i += 256) {
        const float4 d = d4[i], v = p4[i];
        acc(d.x, v.x); acc(d.y, v.y); acc(d.z, v.z); acc(d.w, v.w);
      }
    } else {
      for (int i = threadIdx.x; i < g.HW; i += 256) acc(dy[off + i], x[off + i]);
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    float* o = partial2 + ((int64_t)s * g.C + c) * 2;
    o[0] = s1; o[1] = s2;
  }
}

__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                           const float* __restrict__ partial2, float* __restrict__ dx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, BnGeom g) {
  const int c = blockIdx.x % g.C, s = blockIdx.x / g.C;
  const float mean = mean_in[c], rstd = rstd_in[c], gm = gamma[c], bt = beta[c];
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < g.S; ++k) { s1 += partial2[((int64_t)k * g.C + c) * 2]; s2 += partial2[((int64_t)k * g.C + c) * 2 + 1]; }
  if (s == 0 && threadIdx.x == 0) { dgamma[c] = s2; dbeta[c] = s1; }
  const float inv_n = 1.f / ((float)g.batch * g.HW);
  const float m1 = s1 * inv_n, m2 = s2 * inv_n, k = gm * rstd;
  const int b0 = s * g.nb, b1 = min(b0 + g.nb, g.batch);
  const int hw4 = g.HW >> 2;
  const float sc = rstd * gm, sh = bt - mean * sc;      // the forward's affine: same ReLU mask as bn_apply_fwd_kernel
  auto one = [&](float d, float xv) {
    const float xh = (xv - mean) * rstd;
    const float gg = (relu && fmaf(xv, sc, sh) <= 0.f) ? 0.f : d;
    return k * (gg - m1 - xh * m2);
  };
  for (int b = b0; b < b1; ++b) {
    const int64_t off = ((int64_t)b * g.C + c) * g.HW;
    if ((g.HW & 3) == 0) {
      const float4* d4 = reinterpret_cast<const float4*>(dy + off);
      const float4* p4 = reinterpret_cast<const float4*>(x + off);
      float4* o4 = reinterpret_cast<float4*>(dx + off);
      for (int i = threadIdx.x; i < hw4; i += 256) {
        const float4 d = d4[i], v = p4[i];
        o4[i] = make_float4(one(d.x, v.x), one(d.y, v.y), one(d.z, v.z), one(d.w, v.w));
      }
    } else {
      for (int i = threadIdx.x; i < g.HW; i += 256) dx[off + i] = one(dy[off + i], x[off + i]);
    }
  }
}

// ---- one workgroup per channel, the whole channel in registers (batch * HW <= 512 * VPT): statistics and apply in ONE pass over
// HBM and ONE launch — the 14x14 / 7x7 stages, where the two-kernel form is bound by launch latency (12.7 + 10.4 us for 9.6 MB).
__device__ __forceinline__ float block_sum512(float v, float* red) {     // sum over a 512-thread workgroup, result in every thread
  v = group_sum<16>(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  return ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

template <int VPT>
__global__ __launch_bounds__(512) void bn_fused_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float momentum,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float* __restrict__ y, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, int relu, const float* __restrict__ pre_bias,
                                                           int batch, int C, int HW) {
  __shared__ float red[8];
  const int c = blockIdx.x, n = batch * HW;
  float v[VPT];
  int64_t off[VPT];
  float sum = 0.f;
  DivMod bp(threadIdx.x, 512, HW);                // (image, position) of element e = threadIdx.x + 512 i: one division, not VPT
#pragma unroll
  for (int i = 0; i < VPT; ++i, bp.next()) {
    const int e = threadIdx.x + 512 * i;
    off[i] = e < n ? ((int64_t)bp.q * C + c) * HW + bp.r : -1;
    v[i] = off[i] >= 0 ? x[off[i]] : 0.f;
    sum += v[i];
  }
  const float mean = block_sum512(sum, red) / (float)n;
  float m2 = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) { const float d = off[i] >= 0 ? v[i] - mean : 0.f; m2 = fmaf(d, d, m2); }
  m2 = block_sum512(m2, red);
  const float var = m2 / (float)n;
  const float rstd = 1.0f / sqrtf(var + eps);
  const float sc = rstd * gamma[c], sh = beta[c] - mean * sc;
  if (threadIdx.x == 0) {
    mean_out[c] = mean; rstd_out[c] = rstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (mean + (pre_bias ? pre_bias[c] : 0.f));
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (n > 1 ? m2 / (float)(n - 1) : var);
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i)
    if (off[i] >= 0) { const float o = fmaf(v[i], sc, sh); y[off[i]] = relu ? fmaxf(o, 0.f) : o; }
}

// dxsum (optional, C): per-channel sum of the dx written here = the bias gradient of the convolution that produced x
template <int VPT>
__global__ __launch_bounds__(512) void bn_fused_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                           float* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, float* __restrict__ dxsum, int batch, int C, int HW) {
  __shared__ float red[8];
  const int c = blockIdx.x, n = batch * HW;
  const float mean = mean_in[c], rstd = rstd_in[c], gm = gamma[c], bt = beta[c];
  const float sc = rstd * gm, sh = bt - mean * sc;       // the forward's affine: same ReLU mask as the forward kernels
  float gg[VPT], xh[VPT];
  int64_t off[VPT];
  float s1 = 0.f, s2 = 0.f;
  DivMod bp(threadIdx.x, 512, HW);
#pragma unroll
  for (int i = 0; i < VPT; ++i, bp.next()) {
    const int e = threadIdx.x + 512 * i;
    off[i] = e < n ? ((int64_t)bp.q * C + c) * HW + bp.r : -1;
    const float xv = off[i] >= 0 ? x[off[i]] : 0.f;
    const float d = off[i] >= 0 ? dy[off[i]] : 0.f;
    xh[i] = (xv - mean) * rstd;
    gg[i] = (off[i] < 0 || (relu && fmaf(xv, sc, sh) <= 0.f)) ? 0.f : d;
    s1 += gg[i]; s2 = fmaf(gg[i], xh[i], s2);
  }
  s1 = block_sum512(s1, red);
  s2 = block_sum512(s2, red);
  const float inv_n = 1.f / (float)n;
  const float m1 = s1 * inv_n, m2 = s2 * inv_n, k = gm * rstd;
  float ds = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i)
    if (off[i] >= 0) { const float o = k * (gg[i] - m1 - xh[i] * m2); dx[off[i]] = o; ds += o; }
  if (dxsum != nullptr) ds = block_sum512(ds, red);
  if (threadIdx.x == 0) { dgamma[c] = s2; dbeta[c] = s1; if (dxsum != nullptr) dxsum[c] = ds; }
}

inline int bn_fused_vpt(int batch, int C, int HW) {        // 0: the two-kernel form
  static const int enabled = [] { const char* e = getenv("MM_BN_FUSED"); return e ? atoi(e) : 1; }();   // A/B switch
  const int64_t n = (int64_t)batch * HW;
  if (!enabled || C < 64 || n > 512 * 32) return 0;
  // slots per thread: 7 / 25 are the 7x7 and 14x14 stages at 64 images (3136 and 12544 values per channel) without idle slots
  return n <= 512 * 7 ? 7 : n <= 512 * 8 ? 8 : n <= 512 * 16 ? 16 : n <= 512 * 25 ? 25 : 32;
}

inline BnGeom bn_geom(int batch, int C, int HW) {
  // batch slices: enough workgroups to fill the chip (>= ~1024), at least ~16 KB of a channel per workgroup
  int S = (1024 + C - 1) / C;
  if (S > batch) S = batch;
  const int64_t per_img = (int64_t)HW * 4;
  while (S > 1 && (int64_t)((batch + S - 1) / S) * per_img < 16 * 1024) --S;
  if (S < 1) S = 1;
  BnGeom g;
  g.batch = batch; g.C = C; g.HW = HW; g.nb = (batch + S - 1) / S; g.S = (batch + g.nb - 1) / g.nb; g.SP = g.S;
  return g;
}
}  // namespace

extern "C" {

int mm_bn_splits(int batch, int C, int HW) { return (batch > 0 && C > 0 && HW > 0) ? bn_geom(batch, C, HW).S : 0; }

int mm_bn_relu_fwd(const float* x, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                   float* running_var, float* y, float* mean, float* rstd, float* ws, const float* pre_bias, int relu, int batch, int C,
                   int HW, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || !ws) return MM_ERR_NULL;
  if (batch <= 0 || C <= 0 || HW <= 0) return MM_ERR_SHAPE;
  if ((int64_t)batch * HW < 2) return MM_ERR_SHAPE;          // (torch raises for a single value per channel as well)
  hipStream_t s = (hipStream_t)stream;
  switch (bn_fused_vpt(batch, C, HW)) {
#define MM_BN_FF(V) hipLaunchKernelGGL(bn_fused_fwd_kernel<V>, dim3(C), dim3(512), 0, s, x, gamma, beta, eps, momentum, running_mean, \
                                       running_var, y, mean, rstd, relu, pre_bias, batch, C, HW); return (int)hipGetLastError()
    case 7: MM_BN_FF(7);
    case 8: MM_BN_FF(8);
    case 16: MM_BN_FF(16);
    case 25: MM_BN_FF(25);
    case 32: MM_BN_FF(32);
#undef MM_BN_FF
    default: break;
  }
  const BnGeom g = bn_geom(batch, C, HW);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(C * g.S), dim3(256), 0, s, x, ws, g);
  hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(C * g.S), dim3(256), 0, s, x, ws, gamma, beta, eps, momentum, running_mean, running_var, y, mean,
                     rstd, relu, pre_bias, g);
  return (int)hipGetLastError();
}

int mm_bn_relu_fwd_stats(const float* x, const float* partials, int nparts, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, float* y, float* mean, float* rstd, int relu,
                         int batch, int C, int HW, void* stream) {
  if (!x || !partials || !gamma || !beta || !y || !mean || !rstd) return MM_ERR_NULL;
  if (batch <= 0 || C <= 0 || HW <= 0 || nparts <= 0) return MM_ERR_SHAPE;
  if ((int64_t)batch * HW < 2) return MM_ERR_SHAPE;
  BnGeom g = bn_geom(batch, C, HW);
  g.SP = nparts;
  hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(C * g.S), dim3(256), 0, (hipStream_t)stream, x, partials, gamma, beta, eps, momentum,
                     running_mean, running_var, y, mean, rstd, relu, (const float*)nullptr, g);
  return (int)hipGetLastError();
}

int mm_bn_fused(int batch, int C, int HW) { return (batch > 0 && C > 0 && HW > 0 && bn_fused_vpt(batch, C, HW) > 0) ? 1 : 0; }

int mm_bn_relu_bwd(const float* dy, const float* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                   float* dx, float* dgamma, float* dbeta, float* ws, float* dxsum, int relu, int batch, int C, int HW, void* stream) {
  if (!dy || !x || !gamma || !beta || !mean || !rstd || !dx || !dgamma || !dbeta || !ws) return MM_ERR_NULL;
  if (batch <= 0 || C <= 0 || HW <= 0) return MM_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int vpt = bn_fused_vpt(batch, C, HW);
  if (dxsum != nullptr && vpt == 0) return MM_ERR_UNSUPPORTED;     // only the one-kernel form produces it (mm_bn_fused)
  switch (vpt) {
#define MM_BN_FB(V) hipLaunchKernelGGL(bn_fused_bwd_kernel<V>, dim3(C), dim3(512), 0, s, dy, x, mean, rstd, gamma, beta, relu, dx, dgamma, \
                                       dbeta, dxsum, batch, C, HW); return (int)hipGetLastError()
    case 7: MM_BN_FB(7);
    case 8: MM_BN_FB(8);
    case 16: MM_BN_FB(16);
    case 25: MM_BN_FB(25);
    case 32: MM_BN_FB(32);
#undef MM_BN_FB
    default: break;
  }
  const BnGeom g = bn_geom(batch, C, HW);
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(C * g.S), dim3(256), 0, s, dy, x, mean, rstd, gamma, beta, relu, ws, g);
  hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(C * g.S), dim3(256), 0, s, dy, x, mean, rstd, gamma, beta, relu, ws, dx, dgamma, dbeta, g);
  return (int)hipGetLastError();
}

}  // extern "C"
