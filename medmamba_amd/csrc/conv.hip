// Dense 3x3 convolution (pad 1, stride 1) of the conv branch of SS_Conv_SSM (MedMamba.py:339, 342: nn.Conv2d(C/2, C/2, 3, 1, 1)),
// forward, fp32 on the matrix cores (v_mfma_f32_32x32x2_f32), as an implicit GEMM on NCHW tensors:
//     y[b,k,h,w] = bias[k] + sum_{c,r,s} w[k,c,r,s] * x'[b,c,h+r-1,w+s-1],      x' = relu?(x * scale[c] + shift[c]) inside the image, 0 outside
//     M = K output channels, N = batch*H*W output positions (flattened over the batch: no padded tile slots for 14x14 / 7x7 planes),
//     Kdim = C*9.
// What MIOpen's path costs around its kernel on these shapes (igemm_fwd 83 us at 64 x 192 x 14 x 14): NCHW<->NHWC batched
// transposes, a SubTensorOp, a separate bias add, and the BatchNorm statistics pass over the output that follows in the block
// (bn_stats_kernel, 14 us).  Here the bias is added and the per-channel batch statistics of the output are produced in the
// epilogue, as (count, mean, M2) partials per position tile in the format csrc/bn.hip merges (exact two-pass inside a tile).
//
// Workgroup = 4 wavefronts = 64 output channels x 128 positions; wave (wk, wn) owns 32 channels x 64 positions = two 32x32
// accumulator blocks that share the A operand.  Input channels in chunks of 8 (Kdim chunk 72):
//   staging : every thread owns ONE position of the tile and gathers its 4 channels x 9 taps (zero outside the image, optional
//             per-channel affine + ReLU = a folded BatchNorm in front of the conv) -> LDS im2col tile sX[kappa][n]; the weight
//             tile sW[kappa][k] (row stride 65: conflict-free transposing writes).  The next chunk's values are loaded into
//             registers while the current chunk multiplies.
//   multiply: per kappa pair one ds_read_b32 for A (weights) and two for B (im2col), two MFMAs.
//   epilogue: accumulators -> LDS [k][n] -> (+bias) coalesced NCHW stores and the per-channel tile statistics.
#include "mm_common.h"
#include "medmamba_hip.h"

namespace {
using namespace mm;

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int CV_NT = 128;      // positions per workgroup
constexpr int CV_KT = 64;       // output channels per workgroup
constexpr int CV_CK = 8;        // input channels per chunk
constexpr int CV_KD = CV_CK * 9;   // Kdim per chunk = 72
constexpr int CV_WS = CV_KT + 1;   // sW row stride
constexpr int CV_YS = CV_NT + 1;   // sY row stride (epilogue)

struct ConvParams {
  const float* __restrict__ x;
  const float* __restrict__ w;
  const float* __restrict__ bias;
  const float* __restrict__ aff;     // [scale C | shift C] or nullptr
  float* __restrict__ y;
  float* __restrict__ stats;         // (ntiles, K, 3) or nullptr
  int batch, C, K, H, W, HW, N, ntiles, relu;
};

__global__ __launch_bounds__(256) void conv3x3_fwd_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;                         // [72][128]
  float* sW = smem + CV_KD * CV_NT;         // [72][65]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int tile = blockIdx.x, k0 = blockIdx.y * CV_KT;
  const int wk = wave & 1, wn = wave >> 1;

  // ---- staging identity: one position per thread, 4 of the chunk's 8 channels
  const int nl = t & (CV_NT - 1), half = t >> 7;
  const int n = tile * CV_NT + nl;
  const bool nvalid = n < p.N;
  const int b = nvalid ? n / p.HW : 0, pos = nvalid ? n - b * p.HW : 0;
  const int h = pos / p.W, wq = pos - h * p.W;
  unsigned tapmask = 0;
  int tapoff[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int hh = h + r - 1, ww = wq + s - 1;
      const bool ok = nvalid && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W;
      tapmask |= (ok ? 1u : 0u) << (r * 3 + s);
      tapoff[r * 3 + s] = ok ? (r - 1) * p.W + (s - 1) : 0;
    }
  // All global reads go through bounds-checked buffer descriptors with an out-of-range offset for masked elements (padding taps,
  // channel / position tails): no divergent branch per load — the 54 loads of a chunk are issued back to back and stay in flight
  // while the previous chunk multiplies (guarded plain loads compiled to one exec-masked branch each and serialised).
  const rsrc_t rx = make_rsrc(p.x, (int64_t)p.batch * p.C * p.HW * 4);
  const rsrc_t rw = make_rsrc(p.w, (int64_t)p.K * p.C * 9 * 4);
  const int xoff0 = (b * p.C * p.HW + pos) * 4;            // byte offset of (b, channel 0, pos)
  int xoff[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) xoff[j] = ((tapmask >> j) & 1) ? xoff0 + tapoff[j] * 4 : kOOB;

  float xr[4][9], wr[18];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const int c = c0 + half * 4 + ci;
      const bool cok = c < p.C;
      const int coff = c * p.HW * 4;
      float sc = 1.f, sh = 0.f;
      if (p.aff != nullptr) { sc = p.aff[cok ? c : 0]; sh = p.aff[p.C + (cok ? c : 0)]; }
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const bool ok = cok && ((tapmask >> j) & 1);
        float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ok ? xoff[j] + coff : kOOB, 0, 0));
        if (p.aff != nullptr) {
          v = fmaf(v, sc, sh);
          if (p.relu) v = fmaxf(v, 0.f);
          v = ok ? v : 0.f;                  // the padding is zero AFTER the affine (it pads the BatchNorm's output)
        }
        xr[ci][j] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const int e = t + 256 * i;             // element of the 64 x 72 weight tile, kappa fastest (contiguous in memory)
      const int kl = e / CV_KD, kap = e - kl * CV_KD;
      const int k = k0 + kl, c = c0 + kap / 9;
      wr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, (k < p.K && c < p.C) ? ((k * p.C + c0) * 9 + kap) * 4 : kOOB, 0, 0));
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int j = 0; j < 9; ++j) sX[((half * 4 + ci) * 9 + j) * CV_NT + nl] = xr[ci][j];
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const int e = t + 256 * i;
      const int kl = e / CV_KD, kap = e - kl * CV_KD;
      sW[kap * CV_WS + kl] = wr[i];
    }
  };

  v16f acc0 = {0.f}, acc1 = {0.f};
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  const float* aW = sW + (lane >> 5) * CV_WS + wk * 32 + (lane & 31);
  const float* aX = sX + (lane >> 5) * CV_NT + wn * 64 + (lane & 31);

  const int nchunks = (p.C + CV_CK - 1) / CV_CK;
  load_chunk(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    store_chunk();
    __syncthreads();
    if (ch + 1 < nchunks) load_chunk((ch + 1) * CV_CK);
#pragma unroll 12
    for (int kp = 0; kp < CV_KD / 2; ++kp) {
      const float a = aW[2 * kp * CV_WS];
      const float b0 = aX[2 * kp * CV_NT], b1 = aX[2 * kp * CV_NT + 32];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS [k][n]
  float* sY = smem;                           // [64][129] (the staging tiles are dead)
  {
    const int j = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r >> 2) * 8 + hi * 4 + (r & 3);
      sY[(wk * 32 + i) * CV_YS + wn * 64 + j] = acc0[r];
      sY[(wk * 32 + i) * CV_YS + wn * 64 + 32 + j] = acc1[r];
    }
  }
  __syncthreads();
  // (+bias) NCHW stores: thread = (position, half of the 64 channels)
  {
    float* yb = p.y + (int64_t)b * p.K * p.HW + pos;
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      const int kl = half * 32 + kk, k = k0 + kl;
      if (nvalid && k < p.K) yb[(int64_t)k * p.HW] = sY[kl * CV_YS + nl] + (p.bias ? p.bias[k] : 0.f);
    }
  }
  // per-channel statistics of this tile's outputs (bias included): 4 threads per channel row, exact two-pass
  if (p.stats != nullptr) {
    const int kl = t >> 2, part = t & 3, k = k0 + kl;
    const int nv = min(CV_NT, p.N - tile * CV_NT);        // valid positions of the tile
    const float bk = (p.bias && k < p.K) ? p.bias[k] : 0.f;
    float s = 0.f;
    for (int i = part; i < nv; i += 4) s += sY[kl * CV_YS + i];
    s += dpp_f<DPP_QUAD_XOR1>(s); s += dpp_f<DPP_QUAD_XOR2>(s);
    const float mean = s / (float)nv;
    float m2 = 0.f;
    for (int i = part; i < nv; i += 4) { const float d = sY[kl * CV_YS + i] - mean; m2 = fmaf(d, d, m2); }
    m2 += dpp_f<DPP_QUAD_XOR1>(m2); m2 += dpp_f<DPP_QUAD_XOR2>(m2);
    if (part == 0 && k < p.K) {
      float* o = p.stats + ((int64_t)tile * p.K + k) * 3;
      o[0] = (float)nv; o[1] = mean + bk; o[2] = m2;
    }
  }
}
}  // namespace

extern "C" {

int mm_conv3x3_fwd_tiles(int batch, int H, int W) {
  if (batch <= 0 || H <= 0 || W <= 0) return 0;
  const int64_t N = (int64_t)batch * H * W;
  return (int)((N + CV_NT - 1) / CV_NT);
}

int mm_conv3x3_fwd(const float* x, const float* w, const float* bias, const float* in_affine, int in_relu, float* y, float* stats,
                   int batch, int C, int K, int H, int W, void* stream) {
  if (!x || !w || !y) return MM_ERR_NULL;
  if (batch <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0) return MM_ERR_SHAPE;
  if ((int64_t)batch * C * H * W * 4 >= 0x7ffffff0ll || (int64_t)batch * K * H * W * 4 >= 0x7ffffff0ll || (int64_t)K * C * 36 >= 0x7ffffff0ll)
    return MM_ERR_UNSUPPORTED;      // 32-bit byte offsets inside the descriptors
  ConvParams p;
  p.x = x; p.w = w; p.bias = bias; p.aff = in_affine; p.y = y; p.stats = stats;
  p.batch = batch; p.C = C; p.K = K; p.H = H; p.W = W; p.HW = H * W; p.N = batch * H * W;
  p.ntiles = mm_conv3x3_fwd_tiles(batch, H, W); p.relu = in_relu;
  const size_t lds = sizeof(float) * (size_t)(CV_KD * CV_NT + CV_KD * CV_WS);      // 54.3 KB: below the 64 KB that need no opt-in (>= the epilogue's 64 x 129 floats)
  hipLaunchKernelGGL(conv3x3_fwd_kernel, dim3(p.ntiles, (K + CV_KT - 1) / CV_KT), dim3(256), lds, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

}  // extern "C"
