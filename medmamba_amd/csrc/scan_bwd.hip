// Selective-scan backward for gfx950 (MI355X).  Replaces mamba_ssm's selective_scan_cuda.bwd behind
// SelectiveScanFn.backward (autograd of the call at MedMamba.py:273-279; the adjoint of temp.py:57-139).
//
// Mapping (wave64): workgroup = (batch, direction-group) x CW channels; wavefront = 16 channels;
//   lane = (state group g = lane / 16, channel c = lane % 16); each lane carries 4 of the 16 states.
//   -> sums over the 16 channels of a wave (dB, dC) are DPP row reductions,
//      sums over the 4 state groups (du, ddelta) are two cross-row exchanges.
// Per tile of 64 steps (processed last tile first), per sub-tile of 16 steps (last first):
//   reload the state checkpoint the forward kernel saved at the sub-tile start (x_chk, every 16 steps),
//   recompute the 16 states x_t into registers, then run the adjoint recurrence backwards:
//     gx_t = C_t g_t + a_{t+1} gx_{t+1};  dC_t += g_t x_t;  dB_t += gx_t dl_t u_t;
//     ddl_t = sum_n gx_t (x_{t-1} a_t A_n + B_t u_t);  du_t = sum_n gx_t dl_t B_t + D g_t;
//     dA_n += gx_t x_{t-1} a_t dl_t
//   dB/dC: wave-reduced in registers, accumulated across the workgroup's waves in LDS (ds_add_f32),
//   stored once per tile (plain store when the workgroup owns the whole group, atomics otherwise).
#include "mm_common.h"
#include "medmamba_hip.h"

namespace {
using namespace mm;

struct BwdParams {
  const float* __restrict__ u;
  const float* __restrict__ delta;
  const float* __restrict__ A;
  const float* __restrict__ B;
  const float* __restrict__ C;
  const float* __restrict__ D;
  const float* __restrict__ bias;
  const float* __restrict__ x_chk;
  const float* __restrict__ dout;
  float* __restrict__ du;
  float* __restrict__ ddelta;
  float* __restrict__ dA;
  float* __restrict__ dB;
  float* __restrict__ dC;
  float* __restrict__ dD;
  float* __restrict__ dbias;
  int64_t u_sb, u_sd, d_sb, d_sd, B_sb, B_sg, B_sn, C_sb, C_sg, C_sn;
  int dim, L, G, H, CW, ncw, ntiles, nchk, softplus;
  int ug;                   // channel blocks in u / dout
  unsigned u_map, rev_mask; // block of group g = (u_map >> 4g) & 15; bit g of rev_mask: group g runs backwards
};

// Time-ordered quad (time slots t..t+3) of the row at `row`; rev: slot t is memory position L-1-t.
template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ row, int t, int L, bool rev = false) {
  if constexpr (VEC) {
    if (t >= L) return make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v = *reinterpret_cast<const float4*>(row + (rev ? L - 4 - t : t));
    return rev ? make_float4(v.w, v.z, v.y, v.x) : v;
  } else {
    float4 v;
    v.x = t + 0 < L ? row[rev ? L - 1 - t : t] : 0.f;
    v.y = t + 1 < L ? row[rev ? L - 2 - t : t + 1] : 0.f;
    v.z = t + 2 < L ? row[rev ? L - 3 - t : t + 2] : 0.f;
    v.w = t + 3 < L ? row[rev ? L - 4 - t : t + 3] : 0.f;
    return v;
  }
}
template <bool VEC>
__device__ __forceinline__ void store4(float* __restrict__ row, int t, int L, float4 v, bool rev = false) {
  if constexpr (VEC) {
    if (t < L) *reinterpret_cast<float4*>(row + (rev ? L - 4 - t : t)) = rev ? make_float4(v.w, v.z, v.y, v.x) : v;
  } else {
    if (t + 0 < L) row[rev ? L - 1 - t : t] = v.x;
    if (t + 1 < L) row[rev ? L - 2 - t : t + 1] = v.y;
    if (t + 2 < L) row[rev ? L - 3 - t : t + 2] = v.z;
    if (t + 3 < L) row[rev ? L - 4 - t : t + 3] = v.w;
  }
}
__device__ __forceinline__ float f4get(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

constexpr int NS = 4;    // states per lane
constexpr int CH = 16;   // channels per wave
constexpr int NLD = 4;   // float4 row-loads per lane per tensor per tile

template <bool VEC>
__global__ __launch_bounds__(512) void scan_bwd_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  const int cw = blockIdx.x % p.ncw;
  const int bk = blockIdx.x / p.ncw;
  const int grp = bk % p.G, b = bk / p.G;
  const int ugrp = p.ug < p.G ? (int)((p.u_map >> (4 * grp)) & 15) : grp;
  const bool rev = grp < 32 && ((p.rev_mask >> grp) & 1);

  float* sBC = smem;                                   // [2][16][kTileStride]   B, C tile
  float* sAcc = smem + 2 * kNState * kTileStride;      // [2][16][kTileStride]   dB, dC accumulators
  float* wl = smem + 4 * kNState * kTileStride + wave * (3 * CH * kTileStride);
  float* s_u = wl;                                     // u      -> du   (in place)
  float* s_dl = wl + CH * kTileStride;                 // delta' -> ddl  (in place)
  float* s_g = wl + 2 * CH * kTileStride;              // dout

  // ---- recurrence identity
  const int g = lane >> 4, c = lane & 15;
  const int hc = cw * p.CW + wave * CH + c;
  const bool cvalid = hc < p.H;
  const int d = grp * p.H + (cvalid ? hc : 0);
  float An[NS], A2[NS], dAacc[NS], gx[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    An[j] = p.A[(int64_t)d * kNState + g * NS + j];
    A2[j] = An[j] * kLog2e;
    dAacc[j] = 0.f;
    gx[j] = 0.f;
  }
  const float Dc = p.D ? p.D[d] : 0.f;

  // ---- staging identity
  const int r = lane >> 4, q = lane & 15;
  int64_t uoff[NLD], doff[NLD], ooff[NLD], goff[NLD];
  float bv[NLD], dDacc[NLD], dbacc[NLD];
  bool rvalid[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int hcc = cw * p.CW + wave * CH + 4 * i + r;
    rvalid[i] = hcc < p.H;
    const int dd = grp * p.H + (rvalid[i] ? hcc : 0);
    const int ddu = ugrp * p.H + (rvalid[i] ? hcc : 0);          // channel inside the shared u / dout blocks
    uoff[i] = b * p.u_sb + ddu * p.u_sd;
    doff[i] = b * p.d_sb + dd * p.d_sd;
    ooff[i] = ((int64_t)b * p.dim + dd) * p.L;
    goff[i] = ((int64_t)b * p.ug * p.H + ddu) * p.L;
    bv[i] = p.bias ? p.bias[dd] : 0.f;
    dDacc[i] = 0.f;
    dbacc[i] = 0.f;
  }
  const float* Bbase = p.B + b * p.B_sb + grp * p.B_sg;
  const float* Cbase = p.C + b * p.C_sb + grp * p.C_sg;
  float* dBbase = p.dB + ((int64_t)b * p.G + grp) * kNState * p.L;
  float* dCbase = p.dC + ((int64_t)b * p.G + grp) * kNState * p.L;

  for (int tile = p.ntiles - 1; tile >= 0; --tile) {
    const int t0 = tile * kTile;
    float4 sig[NLD];
    // ---- phase 1: global -> LDS
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int t = t0 + 4 * q;
      const bool ok = rvalid[i];
      const float4 zu = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 vu = ok ? load4<VEC>(p.u + uoff[i], t, p.L, rev) : zu;
      const float4 vd = ok ? load4<VEC>(p.delta + doff[i], t, p.L, rev) : zu;
      const float4 vg = ok ? load4<VEC>(p.dout + goff[i], t, p.L, rev) : zu;
      float4 dl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float raw = f4get(vd, e) + bv[i];
        const bool in = ok && (t + e < p.L);
        float v = p.softplus ? softplus_f(raw) : raw;
        (&dl.x)[e] = in ? v : 0.f;
        (&sig[i].x)[e] = p.softplus ? (raw > 20.f ? 1.f : sigmoid_f(raw)) : 1.f;
        dDacc[i] += f4get(vg, e) * f4get(vu, e);
      }
      const int off = (4 * i + r) * kTileStride + 4 * q;
      *reinterpret_cast<float4*>(s_u + off) = vu;
      *reinterpret_cast<float4*>(s_dl + off) = dl;
      *reinterpret_cast<float4*>(s_g + off) = vg;
    }
    for (int idx = tid; idx < 512; idx += nthreads) {
      const int which = idx >> 8, n = (idx >> 4) & 15, qq = idx & 15;
      const float* src = which ? Cbase + n * p.C_sn : Bbase + n * p.B_sn;
      const int o = (which * kNState + n) * kTileStride + 4 * qq;
      *reinterpret_cast<float4*>(sBC + o) = load4<VEC>(src, t0 + 4 * qq, p.L, rev);
      *reinterpret_cast<float4*>(sAcc + o) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    // ---- phase 2: sub-tiles of 16 steps, last first
    const int tlen = min(kTile, p.L - t0);
    const int nsub = (tlen + kChunk - 1) / kChunk;
    const float* sB = sBC + (g * NS) * kTileStride;
    const float* sC = sBC + (kNState + g * NS) * kTileStride;
    for (int sub = nsub - 1; sub >= 0; --sub) {
      const int ts = sub * kChunk;                  // offset inside the tile
      const int chunk = (t0 >> 4) + sub;            // global chunk index
      float xs[kChunk][NS], x0[NS];
      if (chunk > 0 && cvalid) {
        const float4 v = *reinterpret_cast<const float4*>(
            p.x_chk + (((int64_t)b * p.dim + d) * p.nchk + (chunk - 1)) * kNState + g * NS);
        x0[0] = v.x; x0[1] = v.y; x0[2] = v.z; x0[3] = v.w;
      } else {
        x0[0] = x0[1] = x0[2] = x0[3] = 0.f;
      }
      // forward recompute of the 16 states
#pragma unroll
      for (int tq = 0; tq < 4; ++tq) {
        const int to = ts + 4 * tq;
        const float4 dl4 = *reinterpret_cast<const float4*>(s_dl + c * kTileStride + to);
        const float4 u4 = *reinterpret_cast<const float4*>(s_u + c * kTileStride + to);
        float4 Bv[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) Bv[j] = *reinterpret_cast<const float4*>(sB + j * kTileStride + to);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dl = f4get(dl4, e), dlu = dl * f4get(u4, e);
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            const float a = __builtin_amdgcn_exp2f(dl * A2[j]);
            const float prev = (tq == 0 && e == 0) ? x0[j] : xs[4 * tq + e - 1][j];
            xs[4 * tq + e][j] = fmaf(a, prev, dlu * f4get(Bv[j], e));
          }
        }
      }
      // adjoint recurrence, backwards
#pragma unroll
      for (int tq = 3; tq >= 0; --tq) {
        const int to = ts + 4 * tq;
        const float4 dl4 = *reinterpret_cast<const float4*>(s_dl + c * kTileStride + to);
        const float4 u4 = *reinterpret_cast<const float4*>(s_u + c * kTileStride + to);
        const float4 g4 = *reinterpret_cast<const float4*>(s_g + c * kTileStride + to);
        float4 Bv[NS], Cv[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          Bv[j] = *reinterpret_cast<const float4*>(sB + j * kTileStride + to);
          Cv[j] = *reinterpret_cast<const float4*>(sC + j * kTileStride + to);
        }
        float4 du4, ddl4;
#pragma unroll
        for (int e = 3; e >= 0; --e) {
          const int tt = 4 * tq + e;
          const float dl = f4get(dl4, e), ut = f4get(u4, e), gt = f4get(g4, e);
          const float dlu = dl * ut;
          float ddl = 0.f, duu = 0.f;
          float dBv[NS], dCv[NS];
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            const float Bn = f4get(Bv[j], e), Cn = f4get(Cv[j], e);
            const float xprev = tt == 0 ? x0[j] : xs[tt - 1][j];
            const float gxt = fmaf(Cn, gt, gx[j]);
            const float a = __builtin_amdgcn_exp2f(dl * A2[j]);
            dCv[j] = gt * xs[tt][j];
            dBv[j] = gxt * dlu;
            const float t2 = gxt * xprev * a;
            ddl = fmaf(t2, An[j], ddl);
            ddl = fmaf(gxt, Bn * ut, ddl);
            duu = fmaf(gxt, dl * Bn, duu);
            dAacc[j] = fmaf(t2, dl, dAacc[j]);
            gx[j] = a * gxt;
          }
          // sums over the 4 state groups (rows of 16 lanes)
          ddl += __shfl_xor(ddl, 16); ddl += __shfl_xor(ddl, 32);
          duu += __shfl_xor(duu, 16); duu += __shfl_xor(duu, 32);
          (&ddl4.x)[e] = ddl;
          (&du4.x)[e] = fmaf(Dc, gt, duu);
          // sums over the wave's 16 channels, then one LDS atomic per (n, t) per wave
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            const float sb = group_sum<16>(dBv[j]);
            const float sc = group_sum<16>(dCv[j]);
            if (c == 0) {
              atomicAdd(sAcc + (g * NS + j) * kTileStride + to + e, sb);
              atomicAdd(sAcc + (kNState + g * NS + j) * kTileStride + to + e, sc);
            }
          }
        }
        if (g == 0) {
          *reinterpret_cast<float4*>(s_u + c * kTileStride + to) = du4;
          *reinterpret_cast<float4*>(s_dl + c * kTileStride + to) = ddl4;
        }
      }
    }
    __syncthreads();

    // ---- phase 3: LDS -> global
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int t = t0 + 4 * q;
      const int off = (4 * i + r) * kTileStride + 4 * q;
      const float4 vdu = *reinterpret_cast<const float4*>(s_u + off);
      float4 vdd = *reinterpret_cast<const float4*>(s_dl + off);
      vdd.x *= sig[i].x; vdd.y *= sig[i].y; vdd.z *= sig[i].z; vdd.w *= sig[i].w;
      if (rvalid[i]) {
        store4<VEC>(p.du + ooff[i], t, p.L, vdu, rev);
        store4<VEC>(p.ddelta + ooff[i], t, p.L, vdd, rev);
#pragma unroll
        for (int e = 0; e < 4; ++e) dbacc[i] += (t + e < p.L) ? f4get(vdd, e) : 0.f;
      }
    }
    for (int idx = tid; idx < 512; idx += nthreads) {
      const int which = idx >> 8, n = (idx >> 4) & 15, qq = idx & 15;
      const float4 v = *reinterpret_cast<const float4*>(sAcc + (which * kNState + n) * kTileStride + 4 * qq);
      float* dst = (which ? dCbase : dBbase) + (int64_t)n * p.L;
      const int t = t0 + 4 * qq;
      if (p.ncw == 1) {
        store4<VEC>(dst, t, p.L, v, rev);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (t + e < p.L) atomicAdd(dst + (rev ? p.L - 1 - t - e : t + e), f4get(v, e));
      }
    }
    __syncthreads();
  }

  // ---- per-channel parameter gradients: reduce in the wave, one atomic per value
  if (cvalid) {
#pragma unroll
    for (int j = 0; j < NS; ++j) atomicAdd(p.dA + (int64_t)d * kNState + g * NS + j, dAacc[j]);
  }
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const float sD = group_sum<16>(dDacc[i]);
    const float sb = group_sum<16>(dbacc[i]);
    if (q == 0 && rvalid[i]) {
      const int dd = grp * p.H + cw * p.CW + wave * CH + 4 * i + r;
      if (p.dD) atomicAdd(p.dD + dd, sD);
      if (p.dbias) atomicAdd(p.dbias + dd, sb);
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

namespace mm {

int scan_bwd_launch(const mm_scan_args* a, hipStream_t stream) {
  BwdParams p;
  p.u = a->u; p.delta = a->delta; p.A = a->A; p.B = a->B; p.C = a->C; p.D = a->D; p.bias = a->delta_bias;
  p.x_chk = a->x_chk; p.dout = a->dout; p.du = a->du; p.ddelta = a->ddelta; p.dA = a->dA; p.dB = a->dB;
  p.dC = a->dC; p.dD = a->dD; p.dbias = a->ddelta_bias;
  p.u_sb = a->u_sb; p.u_sd = a->u_sd; p.d_sb = a->delta_sb; p.d_sd = a->delta_sd;
  p.B_sb = a->B_sb; p.B_sg = a->B_sg; p.B_sn = a->B_sn; p.C_sb = a->C_sb; p.C_sg = a->C_sg; p.C_sn = a->C_sn;
  p.dim = a->dim; p.L = a->L; p.G = a->G; p.H = a->dim / a->G;
  p.ntiles = (a->L + kTile - 1) / kTile;
  p.nchk = (a->L + kChunk - 1) / kChunk;
  p.softplus = a->delta_softplus;
  const bool shared = a->u_groups > 0 && a->u_groups < a->G;
  p.ug = shared ? a->u_groups : a->G;
  p.u_map = shared ? a->u_map : 0x76543210u;
  p.rev_mask = a->rev_mask;
  const int waves_needed = (p.H + CH - 1) / CH;
  const int ncw0 = (waves_needed + 7) / 8;                   // <= 8 waves (~122 KB LDS) per workgroup
  const int waves = (waves_needed + ncw0 - 1) / ncw0;
  p.CW = waves * CH;
  p.ncw = (p.H + p.CW - 1) / p.CW;
  const int nblocks = a->batch * a->G * p.ncw;
  const size_t lds = sizeof(float) * (4 * kNState * kTileStride + (size_t)waves * 3 * CH * kTileStride);
  const bool vec = (a->L % 4 == 0) && aligned16(a->u) && aligned16(a->delta) && aligned16(a->B) && aligned16(a->C) &&
                   aligned16(a->dout) && aligned16(a->du) && aligned16(a->ddelta) && aligned16(a->dB) &&
                   aligned16(a->dC) && a->u_sb % 4 == 0 && a->u_sd % 4 == 0 && a->delta_sb % 4 == 0 &&
                   a->delta_sd % 4 == 0 && a->B_sb % 4 == 0 && a->B_sg % 4 == 0 && a->B_sn % 4 == 0 &&
                   a->C_sb % 4 == 0 && a->C_sg % 4 == 0 && a->C_sn % 4 == 0;
  if (vec) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)scan_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(scan_bwd_kernel<true>, dim3(nblocks), dim3(waves * 64), lds, stream, p);
  } else {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)scan_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(scan_bwd_kernel<false>, dim3(nblocks), dim3(waves * 64), lds, stream, p);
  }
  return (int)hipGetLastError();
}

}  // namespace mm
