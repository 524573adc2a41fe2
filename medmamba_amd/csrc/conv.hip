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
// EXPERIMENT (correct, slower than MIOpen): compiled only into the experiments build of the library
// (`python -m medmamba_amd.build --experiments` -> lib/libmedmamba_hip_exp.so, -DMM_EXPERIMENTS); the product library does not
// carry these kernels or their entry points.
#ifdef MM_EXPERIMENTS
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

// ---------------------------------------------------------------------------------------------------------------------------
// Second version.  What changed against conv3x3_fwd_kernel and why:
//   * 2352 accumulator blocks of 32x32 over 1024 SIMDs = 2.3 per SIMD: a grid of 64x128 tiles (294 workgroups, also MIOpen's
//     igemm grid) cannot do better than 2 rounds.  Here a workgroup owns 32 channels x 64 positions = 2 blocks and its 4 waves are
//     (block, Kdim half): waves 0,1 accumulate the first half of the input channels, waves 2,3 the second, summed through LDS at
//     the end (fixed order) — 4.6 half-blocks per SIMD, 92 % of the slots busy in the last round.
//   * the input is staged as the raw PATCH (rows G0-1 .. G1+1 of the batch-stacked image rows that the tile's positions touch,
//     W+2 columns with explicit zero halo, plus one all-zero row per channel) instead of an im2col tile: 9x fewer global loads
//     and LDS writes; the im2col addressing happens in the B-operand reads (per lane 12 precomputed row bases = 4 channels x 3 tap
//     rows, the tap column as an immediate offset; taps outside the image point at the zero row).
//   * weights arrive pre-transposed (Kdim, K) (one torch permute per step): 16-B loads and LDS writes, no index division.
//   * one MFMA contracts the tap (c, r, s) of channels c and c+4 of the chunk: the two lane halves differ by constants only.
constexpr int V2_NT = 64, V2_KT = 32, V2_CK = 8, V2_KD = 72;

struct Conv2Params {
  const float* __restrict__ x;
  const float* __restrict__ wt;      // (C*9, K): wt[(c*9 + r*3 + s)*K + k]
  const float* __restrict__ bias;
  const float* __restrict__ aff;
  float* __restrict__ y;
  float* __restrict__ stats;
  int batch, C, K, H, W, HW, N, relu;
  int NR, PS;                        // staged rows per channel (without the zero row), floats per channel patch = (NR+1)*(W+2)
};

__global__ __launch_bounds__(256) void conv3x3_v2_kernel(const Conv2Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nb = wave & 1, kh = wave >> 1;            // n-block of 32 positions, Kdim half
  const int tg = t & 127;                             // thread inside its Kdim-half group (2 waves)
  const int tile = blockIdx.x, k0 = blockIdx.y * V2_KT;
  const int W2 = p.W + 2;
  const int GSZ = V2_CK * p.PS + V2_KD * V2_KT;       // floats of LDS per Kdim-half group: patch | weights
  float* sP = smem + kh * GSZ;
  float* sW = sP + V2_CK * p.PS;
  const int n0 = tile * V2_NT;
  const int G0 = n0 / p.W;                             // first batch-stacked image row the tile touches

  // ---- B-operand identity: lane -> position n, its three tap rows in the patch (or the zero row)
  const int n = n0 + nb * 32 + (lane & 31);
  const bool nvalid = n < p.N;
  const int G = nvalid ? n / p.W : G0, wq = nvalid ? n - G * p.W : 0;
  const int h = G % p.H;
  int bbase[4][3];                                     // float offsets inside sP: channel c (+4 for the upper lane half), tap row r
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const bool ok = nvalid && h + r - 1 >= 0 && h + r - 1 < p.H;
    const int row = ok ? G - G0 + r : p.NR;            // staged row 0 = stacked row G0-1
#pragma unroll
    for (int c = 0; c < 4; ++c) bbase[c][r] = (c + 4 * (lane >> 5)) * p.PS + row * W2 + wq;
  }
  const float* aW = sW + (lane >> 5) * (36 * V2_KT) + (lane & 31);

  // ---- staging identity: up to 4 patch slots per thread (NR*(W+2) <= 512), 5 weight quads
  const rsrc_t rx = make_rsrc(p.x, (int64_t)p.batch * p.C * p.HW * 4);
  const rsrc_t rw = make_rsrc(p.wt, (int64_t)p.C * 9 * p.K * 4);
  const int nslots = p.NR * W2;
  const int nsl = (nslots + 127) >> 7;
  int sl_lds[4], sl_glb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tg + 128 * i;
    const int rho = idx / W2, j = idx - rho * W2;
    const int Gs = G0 - 1 + rho;
    const bool inside = idx < nslots && Gs >= 0 && Gs < p.batch * p.H && j >= 1 && j <= p.W;
    const int bs = inside ? Gs / p.H : 0, hs = inside ? Gs - bs * p.H : 0;
    sl_lds[i] = idx < nslots ? idx : -1;
    sl_glb[i] = inside ? ((bs * p.C * p.H + hs) * p.W + (j - 1)) * 4 : kOOB;
  }
  const int c_begin = kh * ((p.C + 15) / 16) * 8;     // this half's channels: [c_begin, c_end), multiples of 8
  const int c_end = kh == 0 ? min(p.C, ((p.C + 15) / 16) * 8) : p.C;
  const int nchunks = (max(c_end - c_begin, 0) + V2_CK - 1) / V2_CK;

  float xr[V2_CK][4];
  float4 wr[5];
  // load_chunk ONLY issues loads (nothing consumes a loaded value here: any use — e.g. the optional input affine — would make
  // the compiler wait for the load right there, in front of the MFMA loop that is supposed to hide its latency); the values
  // are finished (affine, ReLU, padding mask) when store_chunk writes them to LDS one MFMA loop later
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int ci = 0; ci < V2_CK; ++ci) {
      const int c = c0 + ci;
      const int coff = c * p.HW * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < nsl)          // (workgroup-uniform: slots in use for this plane width — 1 at 14x14 / 7x7, 2 at 28x28, 3 at 56x56)
          xr[ci][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (c < c_end && sl_glb[i] != kOOB) ? sl_glb[i] + coff : kOOB, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {                      // 72 rows x 8 quads = 576 quads per group of 128 threads
      const int q = tg + 128 * i;
      const int kap = q >> 3, kq = (q & 7) * 4;
      const int c = c0 + kap / 9;
      const bool ok = q < 576 && c < c_end && k0 + kq < p.K;     // (K % 4 == 0 is required by the host wrapper)
      const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? ((c0 * 9 + kap) * p.K + k0 + kq) * 4 : kOOB, 0, 0);
      const v4f f = __builtin_bit_cast(v4f, v);
      wr[i] = make_float4(f.x, f.y, f.z, f.w);
    }
  };
  auto store_chunk = [&](int c0) {
#pragma unroll
    for (int ci = 0; ci < V2_CK; ++ci) {
      const int c = c0 + ci;
      const bool cok = c < c_end;
      float sc = 1.f, sh = 0.f;
      if (p.aff != nullptr) { sc = p.aff[cok ? c : 0]; sh = p.aff[p.C + (cok ? c : 0)]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i >= nsl) continue;
        float v = xr[ci][i];
        if (p.aff != nullptr) {
          v = fmaf(v, sc, sh);
          if (p.relu) v = fmaxf(v, 0.f);
          v = (cok && sl_glb[i] != kOOB) ? v : 0.f;     // the padding is zero AFTER the affine (it pads the BatchNorm's output)
        }
        // slots beyond the patch go to a scratch word behind it instead of branching around the store
        sP[ci * p.PS + (sl_lds[i] >= 0 ? sl_lds[i] : (p.NR + 1) * W2 - 1)] = (sl_lds[i] >= 0) ? v : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int q = tg + 128 * i;
      if (q < 576) *reinterpret_cast<float4*>(sW + (q >> 3) * V2_KT + (q & 7) * 4) = wr[i];
    }
  };

  // the all-zero row of every channel patch (taps above / below the image, positions beyond N)
  for (int i = tg; i < V2_CK * W2; i += 128) sP[(i / W2) * p.PS + p.NR * W2 + (i % W2)] = 0.f;

  v16f acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  if (nchunks > 0) load_chunk(c_begin);
  const int nch_max = (((p.C + 15) / 16) * 8 + V2_CK - 1) / V2_CK;      // both halves run the same number of barriers
  for (int ch = 0; ch < nch_max; ++ch) {
    if (ch < nchunks) store_chunk(c_begin + ch * V2_CK);
    __syncthreads();
    if (ch + 1 < nchunks) load_chunk(c_begin + (ch + 1) * V2_CK);
    if (ch < nchunks) {
      // operands of the NEXT channel pair (9 taps: 9 A + 9 B registers) are read from LDS while the 9 MFMAs of the current one
      // run — left to itself the compiler emits read, s_waitcnt lgkmcnt(0), mfma per step and the matrix pipe idles for every
      // LDS round trip (measured: 32 % of the MFMA rate)
      float av[2][9], bv[2][9];
      auto ld = [&](int c, float (&a)[9], float (&b)[9]) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int s_ = 0; s_ < 3; ++s_) {
            a[r * 3 + s_] = aW[(c * 9 + r * 3 + s_) * V2_KT];
            b[r * 3 + s_] = sP[bbase[c][r] + s_];
          }
      };
      ld(0, av[0], bv[0]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c + 1 < 4) ld(c + 1, av[(c + 1) & 1], bv[(c + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 9; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][j], bv[c & 1][j], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }

  // ---- the two Kdim halves meet: waves 2,3 hand their blocks over through LDS, waves 0,1 add (fixed order)
  float* sR = smem;                                    // [2 blocks][16][64]
  if (kh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sR[(nb * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  float* sY = smem + 2 * 16 * 64;                      // [32 k][65]
  if (kh == 0) {
    const int j = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r >> 2) * 8 + hi * 4 + (r & 3);
      sY[i * 65 + nb * 32 + j] = acc[r] + sR[(nb * 16 + r) * 64 + lane];
    }
  }
  __syncthreads();
  // (+bias) NCHW stores: thread = (position of the tile, quarter of the 32 channels)
  {
    const int nl = t & 63, kq = t >> 6;
    const int ns = n0 + nl;
    if (ns < p.N) {
      const int bs = ns / p.HW, ps = ns - bs * p.HW;
      float* yb = p.y + (int64_t)bs * p.K * p.HW + ps;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int kl = kq * 8 + kk, k = k0 + kl;
        if (k < p.K) yb[(int64_t)k * p.HW] = sY[kl * 65 + nl] + (p.bias ? p.bias[k] : 0.f);
      }
    }
  }
  if (p.stats != nullptr) {                            // 8 threads per channel row, exact two-pass
    const int kl = t >> 3, part = t & 7, k = k0 + kl;
    const int nv = min(V2_NT, p.N - n0);
    const float bk = (p.bias && k < p.K) ? p.bias[k] : 0.f;
    float s = 0.f;
    for (int i = part; i < nv; i += 8) s += sY[kl * 65 + i];
    s += dpp_f<DPP_QUAD_XOR1>(s); s += dpp_f<DPP_QUAD_XOR2>(s); s += dpp_f<DPP_ROW_HALF_MIRROR>(s);
    const float mean = s / (float)nv;
    float m2 = 0.f;
    for (int i = part; i < nv; i += 8) { const float d = sY[kl * 65 + i] - mean; m2 = fmaf(d, d, m2); }
    m2 += dpp_f<DPP_QUAD_XOR1>(m2); m2 += dpp_f<DPP_QUAD_XOR2>(m2); m2 += dpp_f<DPP_ROW_HALF_MIRROR>(m2);
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

int mm_conv3x3_v2_tiles(int batch, int H, int W) {
  if (batch <= 0 || H <= 0 || W <= 0) return 0;
  return (int)(((int64_t)batch * H * W + V2_NT - 1) / V2_NT);
}

int mm_conv3x3_v2_fwd(const float* x, const float* wt, const float* bias, const float* in_affine, int in_relu, float* y, float* stats,
                      int batch, int C, int K, int H, int W, void* stream) {
  if (!x || !wt || !y) return MM_ERR_NULL;
  if (batch <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0) return MM_ERR_SHAPE;
  if (K % 4 != 0 || (reinterpret_cast<uintptr_t>(wt) & 15)) return MM_ERR_ALIGN;
  if ((int64_t)batch * C * H * W * 4 >= 0x7ffffff0ll || (int64_t)batch * K * H * W * 4 >= 0x7ffffff0ll || (int64_t)K * C * 36 >= 0x7ffffff0ll)
    return MM_ERR_UNSUPPORTED;
  Conv2Params p;
  p.x = x; p.wt = wt; p.bias = bias; p.aff = in_affine; p.y = y; p.stats = stats;
  p.batch = batch; p.C = C; p.K = K; p.H = H; p.W = W; p.HW = H * W; p.N = batch * H * W; p.relu = in_relu;
  p.NR = (V2_NT - 2) / W + 2 + 2;                     // stacked rows 64 consecutive positions can touch, + one halo row each side
  p.PS = (p.NR + 1) * (W + 2);
  if (p.NR * (W + 2) > 512) return MM_ERR_UNSUPPORTED;   // 4 patch slots per thread (W <= ~120)
  size_t lds = sizeof(float) * 2 * (size_t)(V2_CK * p.PS + V2_KD * V2_KT);
  const size_t lds_epi = sizeof(float) * (2 * 16 * 64 + 32 * 65);
  if (lds < lds_epi) lds = lds_epi;
  if (lds > 64 * 1024) return MM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(conv3x3_v2_kernel, dim3(mm_conv3x3_v2_tiles(batch, H, W), (K + V2_KT - 1) / V2_KT), dim3(256), lds,
                     (hipStream_t)stream, p);
  return (int)hipGetLastError();
}

}  // extern "C"
#endif  // MM_EXPERIMENTS
