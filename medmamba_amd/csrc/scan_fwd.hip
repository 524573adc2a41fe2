// Selective-scan forward for gfx950 (MI355X).  Replaces mamba_ssm's selective_scan_cuda.fwd behind
// selective_scan_fn (reference call site MedMamba.py:273-279; arithmetic temp.py:57-139).
//
// Mapping (wave64):  one workgroup = one (batch, direction-group) pair x a range of CW channels;
//   one wavefront   = CH = 4*NS channels; lane = (channel c = lane / SG, state group g = lane % SG),
//                     SG = 16/NS lanes share a channel, each lane carries NS of the 16 states in VGPRs.
// Data movement per tile of 64 steps:
//   u, delta  : global (rows contiguous along L) --float4, 16 lanes per row--> regs --softplus, delta*u-->
//               wave-private LDS [channel][t] (row stride 68 floats)  --b128 by (c,t..t+3)--> recurrence
//   B, C      : global --float4--> workgroup-shared LDS [n][t], double buffered, one barrier per tile;
//               they are shared by every channel of the direction (temp.py:95-98), so they are read
//               from HBM/L2 once per workgroup instead of once per channel
//   y         : DPP butterfly over the SG lanes of a channel -> LDS (in place of delta*u) -> + D*u ->
//               float4 global store with the same coalesced mapping as the loads
// The next tile's global loads are issued before the current tile's recurrence (register prefetch).
// The recurrence is sequential in registers: per state-step 1 v_exp_f32 + 4 VALU, no parallel-scan
// work inflation.  Tail / padded steps are made identity steps (delta' = 0 -> a = 1, b = 0).
#include "mm_common.h"
#include "medmamba_hip.h"

namespace {
using namespace mm;

struct FwdParams {
  const float* __restrict__ u;
  const float* __restrict__ delta;
  const float* __restrict__ A;
  const float* __restrict__ B;
  const float* __restrict__ C;
  const float* __restrict__ D;
  const float* __restrict__ bias;
  float* __restrict__ out;
  float* __restrict__ x_chk;
  int64_t u_sb, u_sd, d_sb, d_sd, B_sb, B_sg, B_sn, C_sb, C_sg, C_sn;
  int dim, L, G, H;       // H = channels per group
  int CW, ncw;            // channels per workgroup, workgroups per (batch, group)
  int ntiles, nchk;
  int softplus;
};

template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ p, int t, int L) {
  // p points at element t of a row of length L (t is a multiple of 4)
  if constexpr (VEC) {
    return t < L ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    float4 v;
    v.x = t + 0 < L ? p[0] : 0.f;
    v.y = t + 1 < L ? p[1] : 0.f;
    v.z = t + 2 < L ? p[2] : 0.f;
    v.w = t + 3 < L ? p[3] : 0.f;
    return v;
  }
}

template <bool VEC>
__device__ __forceinline__ void store4(float* __restrict__ p, int t, int L, float4 v) {
  if constexpr (VEC) {
    if (t < L) *reinterpret_cast<float4*>(p) = v;
  } else {
    if (t + 0 < L) p[0] = v.x;
    if (t + 1 < L) p[1] = v.y;
    if (t + 2 < L) p[2] = v.z;
    if (t + 3 < L) p[3] = v.w;
  }
}

__device__ __forceinline__ float f4get(const float4& v, int i) {
  return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

// NS: states per lane (1,2,4).  NBC: float4 B/C staging loads per thread.  VEC: L % 4 == 0 and every
// row 16-B aligned.
template <int NS, int NBC, bool VEC>
__global__ __launch_bounds__(768) void scan_fwd_kernel(const FwdParams p) {
  constexpr int SG = kNState / NS;   // lanes per channel
  constexpr int CH = kWave / SG;     // channels per wave (= 4*NS)
  constexpr int NLD = CH / 4;        // float4 row-loads per lane per tensor per tile (= NS)
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x;
  const int cw = blockIdx.x % p.ncw;
  const int bk = blockIdx.x / p.ncw;
  const int grp = bk % p.G, b = bk / p.G;

  float* bc = smem;                                                    // [2][2][16][kTileStride]
  float* wl = smem + 2 * 2 * kNState * kTileStride + wave * (2 * CH * kTileStride);
  float* s_dl = wl;                                                    // [CH][kTileStride]
  float* s_du = wl + CH * kTileStride;                                 // [CH][kTileStride]  (y in place)

  // ---- recurrence-phase identity: lane -> (channel c, state group g)
  const int c = lane / SG, g = lane % SG;
  const int hc = cw * p.CW + wave * CH + c;        // channel within the group
  const bool cvalid = hc < p.H;
  const int d = grp * p.H + (cvalid ? hc : 0);
  float A2[NS], x[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    A2[j] = p.A[(int64_t)d * kNState + g * NS + j] * kLog2e;
    x[j] = 0.f;
  }

  // ---- staging-phase identity: lane -> (row r of a 4-row group, float4 column q)
  const int r = lane >> 4, q = lane & 15;
  // rows handled by this lane are 4*i + r: one base per tensor + a wave-uniform step keeps VGPRs low
  const int hc0 = cw * p.CW + wave * CH + r;
  const float* ubase = p.u + b * p.u_sb + (int64_t)(grp * p.H + hc0) * p.u_sd;
  const float* dbase = p.delta + b * p.d_sb + (int64_t)(grp * p.H + hc0) * p.d_sd;
  float* obase = p.out + ((int64_t)b * p.dim + grp * p.H + hc0) * p.L;
  float Dv[NLD], bv[NLD];
  bool rvalid[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    rvalid[i] = hc0 + 4 * i < p.H;
    const int dd = grp * p.H + (rvalid[i] ? hc0 + 4 * i : 0);
    Dv[i] = p.D ? p.D[dd] : 0.f;
    bv[i] = p.bias ? p.bias[dd] : 0.f;
  }
  const float* Bbase = p.B + b * p.B_sb + grp * p.B_sg;
  const float* Cbase = p.C + b * p.C_sb + grp * p.C_sg;

  float4 pu[NLD], pd[NLD], pbc[NBC];
  auto issue_loads = [&](int t0) {
    const int t = t0 + 4 * q;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const bool ok = rvalid[i];
      pu[i] = ok ? load4<VEC>(ubase + 4 * i * p.u_sd + t, t, p.L) : make_float4(0.f, 0.f, 0.f, 0.f);
      pd[i] = ok ? load4<VEC>(dbase + 4 * i * p.d_sd + t, t, p.L) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < NBC; ++k) {
      const int idx = tid + k * nthreads;            // 0..511: [which][n][q']
      if (idx < 512) {
        const int which = idx >> 8, n = (idx >> 4) & 15, qq = idx & 15;
        const float* src = which ? Cbase + n * p.C_sn : Bbase + n * p.B_sn;
        pbc[k] = load4<VEC>(src + t0 + 4 * qq, t0 + 4 * qq, p.L);
      }
    }
  };

  issue_loads(0);
  for (int tile = 0; tile < p.ntiles; ++tile) {
    const int t0 = tile * kTile;
    const int buf = tile & 1;
    // ---- phase 1: registers -> LDS (delta' and delta'*u), keep D*u for the epilogue
    float4 uD[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int t = t0 + 4 * q;
      float4 dl, du;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float raw = f4get(pd[i], e) + bv[i];
        float v = p.softplus ? softplus_f(raw) : raw;
        v = (rvalid[i] && t + e < p.L) ? v : 0.f;     // identity step outside the sequence
        (&dl.x)[e] = v;
        (&du.x)[e] = v * f4get(pu[i], e);
      }
      uD[i] = make_float4(pu[i].x * Dv[i], pu[i].y * Dv[i], pu[i].z * Dv[i], pu[i].w * Dv[i]);
      const int off = (4 * i + r) * kTileStride + 4 * q;
      *reinterpret_cast<float4*>(s_dl + off) = dl;
      *reinterpret_cast<float4*>(s_du + off) = du;
    }
#pragma unroll
    for (int k = 0; k < NBC; ++k) {
      const int idx = tid + k * nthreads;
      if (idx < 512) {
        const int which = idx >> 8, n = (idx >> 4) & 15, qq = idx & 15;
        *reinterpret_cast<float4*>(bc + ((buf * 2 + which) * kNState + n) * kTileStride + 4 * qq) = pbc[k];
      }
    }
    if (tile + 1 < p.ntiles) issue_loads(t0 + kTile);
    __syncthreads();

    // ---- phase 2: the recurrence over this tile, 4 steps per iteration
    const int tlen = min(kTile, p.L - t0);
    const int ngroups = (tlen + 3) >> 2;
    const float* sB = bc + (buf * 2 + 0) * kNState * kTileStride + (g * NS) * kTileStride;
    const float* sC = bc + (buf * 2 + 1) * kNState * kTileStride + (g * NS) * kTileStride;
    for (int tg = 0; tg < ngroups; ++tg) {
      const float4 dl4 = *reinterpret_cast<const float4*>(s_dl + c * kTileStride + 4 * tg);
      const float4 du4 = *reinterpret_cast<const float4*>(s_du + c * kTileStride + 4 * tg);
      float4 Bv[NS], Cv[NS];
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        Bv[j] = *reinterpret_cast<const float4*>(sB + j * kTileStride + 4 * tg);
        Cv[j] = *reinterpret_cast<const float4*>(sC + j * kTileStride + 4 * tg);
      }
      float4 y4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dl = f4get(dl4, e), du = f4get(du4, e);
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          const float a = __builtin_amdgcn_exp2f(dl * A2[j]);
          x[j] = fmaf(a, x[j], du * f4get(Bv[j], e));
          y = fmaf(x[j], f4get(Cv[j], e), y);
        }
        (&y4.x)[e] = group_sum<SG>(y);
      }
      if (g == 0) *reinterpret_cast<float4*>(s_du + c * kTileStride + 4 * tg) = y4;
      if (p.x_chk != nullptr && ((tg & 3) == 3 || tg == ngroups - 1) && cvalid) {
        const int chunk = (t0 >> 4) + (tg >> 2);
        float* dst = p.x_chk + (((int64_t)b * p.dim + d) * p.nchk + chunk) * kNState + g * NS;
#pragma unroll
        for (int j = 0; j < NS; ++j) dst[j] = x[j];
      }
    }

    // ---- phase 3: y (+ D*u) LDS -> global, same coalesced mapping as the loads
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int t = t0 + 4 * q;
      const float4 y = *reinterpret_cast<const float4*>(s_du + (4 * i + r) * kTileStride + 4 * q);
      if (rvalid[i])
        store4<VEC>(obase + (int64_t)4 * i * p.L + t, t, p.L, make_float4(y.x + uD[i].x, y.y + uD[i].y, y.z + uD[i].z, y.w + uD[i].w));
    }
  }
}

template <int NS, int NBC, bool VEC>
int launch(const FwdParams& p, int nblocks, int waves, hipStream_t stream) {
  constexpr int CH = 4 * NS;
  const size_t lds = sizeof(float) * (2 * 2 * kNState * kTileStride + (size_t)waves * 2 * CH * kTileStride);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)scan_fwd_kernel<NS, NBC, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((scan_fwd_kernel<NS, NBC, VEC>), dim3(nblocks), dim3(waves * 64), lds, stream, p);
  return (int)hipGetLastError();
}

template <int NS, bool VEC>
int launch_nbc(const FwdParams& p, int nblocks, int waves, hipStream_t stream) {
  const int nbc = (512 + waves * 64 - 1) / (waves * 64);
  if (nbc <= 1) return launch<NS, 1, VEC>(p, nblocks, waves, stream);
  if (nbc <= 2) return launch<NS, 2, VEC>(p, nblocks, waves, stream);
  if (nbc <= 4) return launch<NS, 4, VEC>(p, nblocks, waves, stream);
  return launch<NS, 8, VEC>(p, nblocks, waves, stream);
}

template <int NS>
int launch_vec(const FwdParams& p, int nblocks, int waves, bool vec, hipStream_t stream) {
  return vec ? launch_nbc<NS, true>(p, nblocks, waves, stream) : launch_nbc<NS, false>(p, nblocks, waves, stream);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace mm {

// Host-side planning: pick states-per-lane and the channel range per workgroup.
// More lanes per channel (smaller NS) = more wavefronts for short/narrow problems at the price of
// DPP reduction steps; the target is >= 2 waves per SIMD (2048 waves) chip-wide.
int plan_fwd_variant(int batch, int G, int H, int L) {
  (void)L;
  const long seqs = (long)batch * G * H;
  // waves = seqs / (4*NS)
  if (seqs / 16 >= 4096) return 4;
  if (seqs / 8 >= 3072) return 2;
  if (seqs / 16 >= 2048) return 4;
  if (seqs / 8 >= 1024) return 2;
  return 1;
}

int scan_fwd_launch(const mm_scan_args* a, hipStream_t stream) {
  FwdParams p;
  p.u = a->u; p.delta = a->delta; p.A = a->A; p.B = a->B; p.C = a->C; p.D = a->D; p.bias = a->delta_bias;
  p.out = a->out; p.x_chk = a->x_chk;
  p.u_sb = a->u_sb; p.u_sd = a->u_sd; p.d_sb = a->delta_sb; p.d_sd = a->delta_sd;
  p.B_sb = a->B_sb; p.B_sg = a->B_sg; p.B_sn = a->B_sn; p.C_sb = a->C_sb; p.C_sg = a->C_sg; p.C_sn = a->C_sn;
  p.dim = a->dim; p.L = a->L; p.G = a->G; p.H = a->dim / a->G;
  p.ntiles = (a->L + kTile - 1) / kTile;
  p.nchk = (a->L + kChunk - 1) / kChunk;
  p.softplus = a->delta_softplus;

  int ns = a->variant ? a->variant : plan_fwd_variant(a->batch, a->G, p.H, a->L);
  if (ns != 1 && ns != 2 && ns != 4) return MM_ERR_UNSUPPORTED;
  const int CH = 4 * ns;
  // waves per workgroup: cover the whole group if it fits in 12 waves, else the divisor-friendly split
  const int waves_needed = (p.H + CH - 1) / CH;
  int ncw = (waves_needed + 11) / 12;
  int waves = (waves_needed + ncw - 1) / ncw;
  p.CW = waves * CH;
  p.ncw = (p.H + p.CW - 1) / p.CW;
  const int nblocks = a->batch * a->G * p.ncw;

  const bool vec = (a->L % 4 == 0) && aligned16(a->u) && aligned16(a->delta) && aligned16(a->B) && aligned16(a->C) &&
                   aligned16(a->out) && a->u_sb % 4 == 0 && a->u_sd % 4 == 0 && a->delta_sb % 4 == 0 &&
                   a->delta_sd % 4 == 0 && a->B_sb % 4 == 0 && a->B_sg % 4 == 0 && a->B_sn % 4 == 0 &&
                   a->C_sb % 4 == 0 && a->C_sg % 4 == 0 && a->C_sn % 4 == 0;
  switch (ns) {
    case 1: return launch_vec<1>(p, nblocks, waves, vec, stream);
    case 2: return launch_vec<2>(p, nblocks, waves, vec, stream);
    default: return launch_vec<4>(p, nblocks, waves, vec, stream);
  }
}

}  // namespace mm
