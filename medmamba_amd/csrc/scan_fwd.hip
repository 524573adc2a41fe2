// Selective-scan forward for gfx950 (MI355X).  Replaces mamba_ssm's selective_scan_cuda.fwd behind
// selective_scan_fn (reference call site MedMamba.py:273-279; arithmetic temp.py:57-139).
//
// Mapping (wave64): every wavefront is independent (no workgroup barrier anywhere):
//   wavefront = one (batch, direction-group) pair x CH = 4*NS consecutive channels;
//   lane = (channel c = lane / SG, state group g = lane % SG), SG = 16/NS lanes share a channel and each
//   lane carries NS of the 16 states in VGPRs for the whole sequence.
// Data movement per tile of 64 steps, all through wave-private LDS:
//   u, delta : buffer_load_dwordx4, 16 lanes per row (256 B contiguous per row) -> regs -> softplus,
//              delta*u -> LDS [channel][t] (row stride 68 floats) -> ds_read_b128 by (c, t..t+3)
//   B, C     : shared by every channel of the direction (temp.py:95-98): HBM once, then L2 (workgroups of
//              one (batch, direction) are mapped to the same XCD) -> regs -> LDS [n][t]
//   y        : DPP butterfly over the SG lanes of a channel -> LDS (in place of delta*u) -> + D*u ->
//              buffer_store_dwordx4 with the same coalesced mapping as the loads
// Software pipeline: the next tile's global loads are issued before the current tile's recurrence, the
// current tile's stores are issued after the next wait point (loads are older than stores in the vmcnt
// queue, so waiting for the loads never waits for the stores), and inside the recurrence the next 4-step
// group's LDS operands are read while the current group's arithmetic runs.
// All global accesses go through bounds-checked buffer descriptors: out-of-range lanes (sequence tail,
// channel tail) read 0 and drop their stores, so there is no divergent control flow and the compiler can
// count vmcnt exactly.  Padded steps are identity steps (delta' = 0 -> a = 1, b = 0).
// Cost per state-step: v_mul, v_exp_f32, v_mul, v_fma, v_fma (measured: ~1.05 ns per VALU issue slot per
// SIMD with >= 2 waves; the transcendental overlaps other VALU) — no parallel-scan work inflation.
#include <type_traits>
#include "mm_common.h"
#include "medmamba_hip.h"

namespace {
using namespace mm;

constexpr int fwd_tile(int ns, bool lean) { return (lean && ns >= 2) ? 32 : 64; }
// B / C tiles lie in LDS as [position][16 states] rows of kBCS floats (16 + 4 pad: 16-B aligned rows; the rows are ordered by
// quad component first, see the staging stores): the NS states of a lane at one time step are ONE ds_read_b32/b64/b128, and they arrive as
// adjacent registers — operand pairs of v_pk_fma_f32 / v_pk_mul_f32 (two states per instruction)
// (18 for 2 states per lane: 8-B aligned rows suffice for ds_read_b64, and 2 x 64 x 20 floats would push a 2-wave workgroup
// over 160 KB / 6 — the 56x56 stage runs 3072 such waves and needs all 3 per SIMD resident at once)
constexpr int bcs_of(int ns) { return ns == 4 ? 20 : 18; }
typedef float v2f __attribute__((ext_vector_type(2)));

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
  int batch, dim, L, G, H;  // H = channels per group
  int wpg;                  // waves per (batch, group)
  int ug;                   // channel blocks in u
  unsigned u_map, rev_mask; // block of group g = (u_map >> 4g) & 15; bit g of rev_mask: group g runs backwards
  int ntiles, nchk;
  int nwaves_total;
  // fused dt projection (MedMamba.py:262; inference): delta[d,t] = sum_r dtw[d,r] * dts[b,g,r,t] computed while staging
  const float* __restrict__ dtw;   // (dim, R) in this kernel's direction order, or nullptr (then `delta` is read)
  const float* __restrict__ dts;   // (batch, G, R, L) rows with strides (dts_sb, dts_sg, dts_sn, 1)
  int64_t dts_sb, dts_sg, dts_sn;
  int R;
#ifdef MM_EXPERIMENTS
  int dbg;   // experiments build only — timing ablations (results are wrong when set): 1 no y store, 2 no recurrence
#endif
};
// the product library has no ablation bits: the expression folds to 0 and the code they guarded disappears
#ifdef MM_EXPERIMENTS
#define MM_FWD_DBG(p) ((p).dbg)
#else
#define MM_FWD_DBG(p) 0
#endif

__device__ __forceinline__ float f4get(const float4& v, int i) {
  return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

// NS: states per lane (1,2,4).  VEC: L % 4 == 0 and all rows 16-B aligned.  SP: delta_softplus.
// LEAN: register diet for shapes with plenty of wavefronts (>= 3 per SIMD hide latency by themselves): B/C are
//       loaded where they are consumed instead of one tile ahead, the recurrence uses one operand set, and the
//       tile's stores are issued as soon as they exist.  Without LEAN (few, long wavefronts) everything is
//       software-pipelined in registers.
// DT: the dt projection (rank R <= kDtMax) is fused into the staging phase: no (batch, K*D, L) delta tensor is read.
constexpr int kDtMax = 8;
// SH (LEAN, 4 waves per workgroup that all belong to one (batch, direction)): the B / C tile is staged ONCE per workgroup — each
//     wave loads one quarter, one tile ahead — into a double-buffered shared tile, one barrier per tile.  Wave-private staging
//     asked the L1 for every B / C line four times at once (TCP_PENDING_STALL 50-69 % of the kernel, profiles/r4_fwd_mem_counters_*).
template <int NS, bool VEC, bool SP, bool LEAN, bool DT = false, bool SH = false>
__global__ __launch_bounds__(256) void scan_fwd_kernel(const FwdParams p) {
  constexpr int SG = kNState / NS;   // lanes per channel
  constexpr int CH = kWave / SG;     // channels per wave (= 4*NS)
  // tile geometry: LEAN wavefronts use 32-step tiles (half the LDS per wave -> >= 3 waves per SIMD fit)
  constexpr int kTile = fwd_tile(NS, LEAN);
  constexpr int kTileStride = kTile + 4;          // +16 B pad: conflict-free b128 rows for 32 and 64
  constexpr int QL = kTile / 4;                   // lanes (float4 columns) per row
  constexpr int RPI = kWave / QL;                 // rows per load instruction
  constexpr int NLD = CH / RPI;                   // float4 row-loads per lane per tensor per tile
  constexpr int NBC = 2 * kNState / RPI;          // float4 B/C staging loads per lane per tile
  constexpr int NPBC = LEAN ? 1 : NBC;
  constexpr int kBCS = bcs_of(NS);
  constexpr int WPRIV = 2 * CH * kTileStride, BCSZ = 2 * kTile * kBCS;
  constexpr int WLDS = SH ? WPRIV : WPRIV + BCSZ;                          // floats of LDS per wave
  static_assert(!SH || (LEAN && NBC == 4 && !DT), "shared B/C staging: 4 waves x one quarter of a 32-step tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = blockDim.x >> 6;
  // XCD-aware numbering: blocks b and b+8 share an XCD (its L2); give each XCD a contiguous range of
  // logical wave ids so that the waves of one (batch, direction) — which all stream the same B/C — meet in one L2.
  int blk = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) blk = (blk & 7) * (nblk >> 3) + (blk >> 3);
  const int gw = blk * wpb + wave;                   // logical wave id
  if (gw >= p.nwaves_total) return;
  const int cwv = gw % p.wpg;                        // channel tile inside the group
  const int bk = gw / p.wpg;
  const int grp = bk % p.G, b = bk / p.G;
  const int ugrp = p.ug < p.G ? (int)((p.u_map >> (4 * grp)) & 15) : grp;      // which channel block of u this direction reads
  const bool rev = grp < 32 && ((p.rev_mask >> grp) & 1);          // this direction runs over the sequence backwards

  float* wl = smem + wave * WLDS;
  float* s_dl = wl;                                   // [CH][kTileStride]   delta'
  float* s_du = wl + CH * kTileStride;                // [CH][kTileStride]   delta'*u, then y (in place)
  // [2][kTile][kBCS] B, C: row = position inside the tile, column = state.  SH: two such tiles behind the waves' private rows
  float* s_bc0 = SH ? smem + wpb * WPRIV : wl + 2 * CH * kTileStride;

  // ---- recurrence identity: lane -> (channel c, state group g)
  const int c = lane / SG, g = lane % SG;
  const int hc = cwv * CH + c;                        // channel within the group
  const bool cvalid = hc < p.H;
  const int d = grp * p.H + (cvalid ? hc : 0);
  float A2[NS], x[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    A2[j] = p.A[(int64_t)d * kNState + g * NS + j] * kLog2e;
    x[j] = 0.f;
  }

  // ---- staging identity: lane -> (row r of a 4-row group, float4 column q); rows 4*i + r
  const int r = lane / QL, q = lane % QL;
  const int hc0 = cwv * CH + r;
  const int d0 = grp * p.H + cwv * CH;               // first channel of this wave (wave-uniform)
  const int d0u = ugrp * p.H + cwv * CH;             // ... inside u
  // descriptors: wave-uniform bases that cover exactly this wave's rows (so 32-bit offsets suffice for any channel
  // stride, e.g. channel-major planes with stride batch*L); the hardware range check does the tail masking
  const int nrw = min(CH, p.H - cwv * CH);           // valid rows of this wave (<= 0: empty descriptors)
  const rsrc_t ru = make_rsrc(p.u + b * p.u_sb + d0u * p.u_sd, ((int64_t)(nrw - 1) * p.u_sd + p.L) * 4);
  const rsrc_t rd = make_rsrc(p.delta + b * p.d_sb + d0 * p.d_sd, ((int64_t)(nrw - 1) * p.d_sd + p.L) * 4);
  const rsrc_t ro = make_rsrc(p.out + ((int64_t)b * p.dim + d0) * p.L, (int64_t)nrw * p.L * 4);
  const rsrc_t rB = make_rsrc(p.B + b * p.B_sb + grp * p.B_sg, ((int64_t)(kNState - 1) * p.B_sn + p.L) * 4);
  const rsrc_t rC = make_rsrc(p.C + b * p.C_sb + grp * p.C_sg, ((int64_t)(kNState - 1) * p.C_sn + p.L) * 4);
  const rsrc_t rT = make_rsrc(DT ? p.dts + b * p.dts_sb + grp * p.dts_sg : nullptr, DT ? ((int64_t)(p.R - 1) * p.dts_sn + p.L) * 4 : 0);
  const int tstep = DT ? (int)p.dts_sn * 4 : 0;
  float Dv[NLD], bv[NLD];
  bool rvalid[NLD];
  int uoff[NLD], doff[NLD], ooff[NLD];
  float wdt[DT ? NLD : 1][DT ? kDtMax : 1];        // this lane's rows of the dt projection weight
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    rvalid[i] = hc0 + RPI * i < p.H;
    const int dd = grp * p.H + (rvalid[i] ? hc0 + RPI * i : 0);
    Dv[i] = p.D ? p.D[dd] : 0.f;
    bv[i] = p.bias ? p.bias[dd] : 0.f;
    uoff[i] = (int)((r + RPI * i) * p.u_sd) * 4;
    doff[i] = (int)((r + RPI * i) * p.d_sd) * 4;
    ooff[i] = ((r + RPI * i) * p.L) * 4;
    if constexpr (DT) {
#pragma unroll
      for (int k = 0; k < kDtMax; ++k) wdt[i][k] = k < p.R ? p.dtw[(int64_t)dd * p.R + k] : 0.f;
    }
  }
  // B/C staging: NBC float4 per lane per tile: k -> (which = k / (NBC/2), n = (k % (NBC/2)) * RPI + r, column q)
  int bcoff[NBC];
#pragma unroll
  for (int k = 0; k < NBC; ++k) {
    const int n = (k % (NBC / 2)) * RPI + r;
    bcoff[k] = (int)(n * ((k >= NBC / 2) ? p.C_sn : p.B_sn)) * 4;
  }

  // SH: this wave's quarter of the tile = pass `wave` of the staging loop below
  const int n_w = (wave % (NBC / 2)) * RPI + r;
  const bool isC_w = wave >= NBC / 2;
  const int bcoff_w = (int)(n_w * (isC_w ? p.C_sn : p.B_sn)) * 4;
  float4 pu[NLD], pd[DT ? 1 : NLD], pbc[NPBC], pdt[DT ? kDtMax : 1];
  auto issue_loads = [&](int t0) {
    const int t = t0 + 4 * q;
    if constexpr (SH) pbc[0] = load_quad<VEC, VEC>(isC_w ? rC : rB, bcoff_w, t, p.L, rev, true);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      pu[i] = load_quad<VEC, VEC>(ru, uoff[i], t, p.L, rev, rvalid[i]);
      if constexpr (!DT) pd[i] = load_quad<VEC, VEC>(rd, doff[i], t, p.L, rev, rvalid[i]);
    }
    if constexpr (DT) {   // the R rows of dts are shared by every channel of the direction (L1 / L2 after the first wave)
#pragma unroll
      for (int k = 0; k < kDtMax; ++k) pdt[k] = load_quad<VEC, VEC>(rT, k * tstep, t, p.L, rev, k < p.R);
    }
    if constexpr (!LEAN) {
#pragma unroll
      for (int k = 0; k < NBC; ++k) pbc[k] = load_quad<VEC, VEC>((k >= NBC / 2) ? rC : rB, bcoff[k], t, p.L, rev, true);
    }
  };
  // Reversed directions on the vector path: quads stay in MEMORY order all the way (no per-component selects on loads
  // and stores: ~16 v_cndmask per 4-step group saved); a reversed tile lies mirrored in LDS — time quad q in column
  // QL-1-q, time step e of a quad in component 3-e — and the recurrence walks it backwards (instantiated for both orders,
  // selected by a wave-uniform branch).  The dword path (VEC = false) keeps everything in time order.
  const bool revm = VEC && rev;
  const int qc = revm ? QL - 1 - q : q;               // LDS column of this lane's staged quads

  float4 yreg[NLD];
  auto store_tile = [&](int t0) {
    const bool st_en = !(MM_FWD_DBG(p) & 1);   // folded into the range check: no branch, so vmcnt stays countable
#pragma unroll
    for (int i = 0; i < NLD; ++i) store_quad<VEC, VEC>(ro, ooff[i], t0 + 4 * q, p.L, rev, rvalid[i] && st_en, yreg[i]);
  };

  issue_loads(0);
  for (int tile = 0; tile < p.ntiles; ++tile) {
    const int t0 = tile * kTile;
    // ---- phase 1: registers -> LDS (delta' and delta'*u), keep D*u for the epilogue
    float4 uD[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int t = t0 + 4 * q;
      float4 dl, du;
      float4 dr;
      if constexpr (DT) {
        dr = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < kDtMax; ++k) {      // rows k >= R were loaded as zeros with zero weights
          dr.x = fmaf(wdt[i][k], pdt[k].x, dr.x); dr.y = fmaf(wdt[i][k], pdt[k].y, dr.y);
          dr.z = fmaf(wdt[i][k], pdt[k].z, dr.z); dr.w = fmaf(wdt[i][k], pdt[k].w, dr.w);
        }
      } else {
        dr = pd[i];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float raw = f4get(dr, e) + bv[i];
        float v = SP ? softplus_f(raw) : raw;
        // identity step outside the sequence / channel range (vector path: L % 4 == 0, a quad is inside or outside as a whole)
        v = (rvalid[i] && (VEC ? t < p.L : t + e < p.L)) ? v : 0.f;
        (&dl.x)[e] = v;
        (&du.x)[e] = v * f4get(pu[i], e);
      }
      uD[i] = make_float4(pu[i].x * Dv[i], pu[i].y * Dv[i], pu[i].z * Dv[i], pu[i].w * Dv[i]);
      const int off = (RPI * i + r) * kTileStride + 4 * qc;
      *reinterpret_cast<float4*>(s_dl + off) = dl;
      *reinterpret_cast<float4*>(s_du + off) = du;
    }
    float* s_bc = SH ? s_bc0 + (tile & 1) * BCSZ : s_bc0;
    if constexpr (SH) {
      const float4 v = pbc[0];
      float* dst = s_bc + ((isC_w ? kTile : 0) + qc) * kBCS + n_w;
      dst[0] = v.x; dst[QL * kBCS] = v.y; dst[2 * QL * kBCS] = v.z; dst[3 * QL * kBCS] = v.w;
    }
#pragma unroll
    for (int k = 0; k < (SH ? 0 : NBC); ++k) {
      const int n = (k % (NBC / 2)) * RPI + r;
      const bool isC = k >= NBC / 2;
      const float4 v = LEAN ? load_quad<VEC, VEC>(isC ? rC : rB, bcoff[k], t0 + 4 * q, p.L, rev, true) : pbc[LEAN ? 0 : k];
      // 4 positions of state n, transposed into [position][n] rows.  Position p = 4*column + e lies in row e*QL + column (rows
      // grouped by quad component): consecutive lanes of one ds_write_b32 then write CONSECUTIVE rows, 20 (18) floats apart, so
      // the 32 lanes of a bank group spread over >= 16 banks (2-way at most, which a b32 store hides) — with rows in position order
      // the lanes were 4 rows = 80 floats = 16 banks apart and only 4 banks took all 32 stores (8-way)
      float* dst = s_bc + ((isC ? kTile : 0) + qc) * kBCS + n;
      dst[0] = v.x; dst[QL * kBCS] = v.y; dst[2 * QL * kBCS] = v.z; dst[3 * QL * kBCS] = v.w;
    }
    // the previous tile's stores go out here: older than the loads issued next, so the wait for those
    // loads (one recurrence later) retires them for free and every path sees the same vmcnt picture
    if constexpr (!LEAN) {
      if (tile > 0) store_tile(t0 - kTile);
    }
    if (tile + 1 < p.ntiles) issue_loads(t0 + kTile);
    // SH: every quarter of this tile's B / C is in LDS behind this barrier; the other buffer is free again because every wave has
    // finished the previous tile's recurrence before it got here
    if constexpr (SH) __syncthreads();

    // ---- phase 2: the recurrence over this tile, 4 steps per group
    const int tlen = min(kTile, p.L - t0);
    const int ngroups = (MM_FWD_DBG(p) & 2) ? 0 : (tlen + 3) >> 2;
    const float* sB = s_bc + g * NS;                               // this lane's states inside a [t][16] row
    const float* sC = s_bc + kTile * kBCS + g * NS;
    struct Ops { float4 dl4, du4; float Bt[4][NS], Ct[4][NS]; };
    auto phase2 = [&](auto rvtag) {
    constexpr bool RV = decltype(rvtag)::value;          // mirrored tile: time step tau lies at position kTile-1-tau
    auto load_row = [](const float* src, float (&dst)[NS]) {
      if constexpr (NS == 4) { const float4 v = *reinterpret_cast<const float4*>(src); dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
      else if constexpr (NS == 2) { const float2 v = *reinterpret_cast<const float2*>(src); dst[0] = v.x; dst[1] = v.y; }
      else dst[0] = src[0];
    };
    auto load_ops = [&](int tg) {
      Ops o;
      const int col = RV ? QL - 1 - tg : tg;
      o.dl4 = *reinterpret_cast<const float4*>(s_dl + c * kTileStride + 4 * col);
      o.du4 = *reinterpret_cast<const float4*>(s_du + c * kTileStride + 4 * col);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // step e of group tg is position 4*tg + e (mirrored tile: kTile-1-(4*tg+e) = 4*(QL-1-tg) + 3-e) -> row component*QL + column
        const int row = RV ? (3 - e) * QL + (QL - 1 - tg) : e * QL + tg;
        load_row(sB + row * kBCS, o.Bt[e]);
        load_row(sC + row * kBCS, o.Ct[e]);
      }
      return o;
    };
    auto compute = [&](const Ops& o, int tg) {
      auto at = [](const float4& v, int e) { return f4get(v, RV ? 3 - e : e); };
      // all 4*NS decay factors of the group first, then the dependent FMA chains: a v_exp_f32 result that is consumed
      // 2-3 instructions later stalls a wave that has no partner on its SIMD (the transcendental pipe takes 8 cycles
      // per instruction, but it runs beside the VALU) — keeping the two blocks apart removes that exposure.
      // The multiplies and FMAs work on PAIRS of states (v2f -> v_pk_mul_f32 / v_pk_fma_f32, the per-step scalars delta',
      // delta'*u broadcast by op_sel): 2.5 instead of 4 VALU per state-step next to the v_exp_f32.
      float4 y4;
      if constexpr (NS >= 2) {
        constexpr int NP = NS / 2;
        v2f a[4][NP];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int jj = 0; jj < NP; ++jj) {
            const v2f pw = (v2f){A2[2 * jj], A2[2 * jj + 1]} * at(o.dl4, e);
            a[e][jj] = (v2f){__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float du = at(o.du4, e);
          v2f yy = {0.f, 0.f};
#pragma unroll
          for (int jj = 0; jj < NP; ++jj) {
            v2f xx = {x[2 * jj], x[2 * jj + 1]};
            xx = a[e][jj] * xx + (v2f){o.Bt[e][2 * jj], o.Bt[e][2 * jj + 1]} * du;
            yy = xx * (v2f){o.Ct[e][2 * jj], o.Ct[e][2 * jj + 1]} + yy;
            x[2 * jj] = xx.x; x[2 * jj + 1] = xx.y;
          }
          (&y4.x)[RV ? 3 - e : e] = group_sum<SG>(yy.x + yy.y);
        }
      } else {
        float a[4][NS];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < NS; ++j) a[e][j] = __builtin_amdgcn_exp2f(at(o.dl4, e) * A2[j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float du = at(o.du4, e);
          float y = 0.f;
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            x[j] = fmaf(a[e][j], x[j], du * o.Bt[e][j]);
            y = fmaf(x[j], o.Ct[e][j], y);
          }
          (&y4.x)[RV ? 3 - e : e] = group_sum<SG>(y);
        }
      }
      if (g == 0) *reinterpret_cast<float4*>(s_du + c * kTileStride + 4 * (RV ? QL - 1 - tg : tg)) = y4;
      if (p.x_chk != nullptr && ((tg & 3) == 3 || tg == ngroups - 1) && cvalid) {
        const int chunk = (t0 >> 4) + (tg >> 2);   // kChunk = 16 divides both tile sizes
        // layout (batch, chunk, dim, 16): the 16 states of a wave's 4*NS channels are ONE contiguous run (1 KB at NS = 4), for
        // this store and for the backward kernel's load alike ((batch, dim, chunk, 16) scattered 64-B pieces: the training
        // forward ran 45 % slower than the inference forward at the 14x14 stage)
        float* dst = p.x_chk + (((int64_t)b * p.nchk + chunk) * p.dim + d) * kNState + g * NS;
#pragma unroll
        for (int j = 0; j < NS; ++j) dst[j] = x[j];
      }
    };
    // two operand register sets, rotated by hand (a `cur = nxt` copy costs 40 v_mov per group)
    if constexpr (LEAN) {
      for (int tg = 0; tg < ngroups; ++tg) {
        const Ops o = load_ops(tg);
        compute(o, tg);
      }
    } else {
    Ops opA = load_ops(0);
    for (int tg = 0; tg < ngroups; tg += 2) {
      Ops opB = load_ops(min(tg + 1, ngroups - 1));
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch above the arithmetic it overlaps with
      compute(opA, tg);
      __builtin_amdgcn_sched_barrier(0);
      opA = load_ops(min(tg + 2, ngroups - 1));
      __builtin_amdgcn_sched_barrier(0);
      if (tg + 1 < ngroups) compute(opB, tg + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    };   // phase2
    if constexpr (VEC) {
      if (revm) phase2(std::true_type{}); else phase2(std::false_type{});
    } else {
      phase2(std::false_type{});
    }

    // ---- phase 3: y (+ D*u) LDS -> registers -> global (coalesced like the loads)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const float4 y = *reinterpret_cast<const float4*>(s_du + (RPI * i + r) * kTileStride + 4 * qc);
      yreg[i] = make_float4(y.x + uD[i].x, y.y + uD[i].y, y.z + uD[i].z, y.w + uD[i].w);
    }
    if constexpr (LEAN) store_tile(t0);
  }
  if constexpr (!LEAN) store_tile((p.ntiles - 1) * kTile);
}


// ======================================================================================================================
// Workgroup-cooperative form for LONG sequences with FEW sequences (round 4; the 56x56 stage of T / S, the 96x96 stage of B).
// There the general kernel has 1536 wavefronts of 4 states per lane for 1024 SIMDs: half of the SIMDs run two wavefronts (95 ns
// per step for the pair), half run one (62 ns per step, then idle) — and every wavefront stages the whole B / C tile for itself.
// Here a workgroup takes one (batch, direction) (or half of its channels): W wavefronts x 8 channels, lane = (state group
// g = lane / 8, channel c = lane % 8), 2 states per lane -> 3072 wavefronts for that stage, three per SIMD on every SIMD (one
// 12-wave workgroup per CU at 64 images).  What makes 2 states per lane affordable, which it was not in the general kernel:
//   * B / C are staged ONCE per workgroup (plain coalesced b128 stores, [state][position] rows, double buffered: one barrier
//     per 64-step tile) instead of once per wavefront with a transposing store;
//   * the recurrence reads them with ONE ds_read_b32 per state and 4 steps (lane c of a quad holds step c % 4) and the FMAs pick
//     the step through DPP quad_perm:[s,s,s,s]: 1 LDS read per step and wave instead of 2.5;
//   * y: three permlane swaps + one row_ror:8 add per 4 steps sum the 8 state groups.
template <int S> __device__ __forceinline__ void fmac_q(float& acc, float b, float m) {      // acc += (lane S of b's quad) * m
  asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(S));
}
template <int S> __device__ __forceinline__ float mul_q(float b, float m) {
  float r;
  asm("v_mul_f32_dpp %0, %1, %2 quad_perm:[%3,%3,%3,%3] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(b), "v"(m), "n"(S));
  return r;
}
__device__ __forceinline__ void fmac_qsel(int s, float& acc, float b, float m) {
  switch (s) { case 0: fmac_q<0>(acc, b, m); break; case 1: fmac_q<1>(acc, b, m); break; case 2: fmac_q<2>(acc, b, m); break;
               default: fmac_q<3>(acc, b, m); break; }
}
__device__ __forceinline__ float mul_qsel(int s, float b, float m) {
  switch (s) { case 0: return mul_q<0>(b, m); case 1: return mul_q<1>(b, m); case 2: return mul_q<2>(b, m); default: return mul_q<3>(b, m); }
}

constexpr int kWgT = 64, kWgTS = kWgT + 4, kWgCH = 8;
constexpr int kWgBC = 2 * 2 * kNState * kWgTS;                      // floats of the double-buffered B / C tile
constexpr int kWgWave = 2 * kWgCH * kWgTS;                          // floats per wave: delta', delta'*u / y

template <bool SP>
__global__ __launch_bounds__(1024) void scan_fwd_wg_kernel(const FwdParams p) {
  constexpr int CH = kWgCH, kT = kWgT, kTS = kWgTS, QL = kT / 4, RPI = kWave / QL, NLD = CH / RPI;   // 16 lanes per row, 4 rows, 2 loads
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x, nwv = nthreads >> 6;
  int blk = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) blk = (blk & 7) * (nblk >> 3) + (blk >> 3);      // workgroups of one (batch, direction) meet in one XCD's L2
  const int cwb = blk % p.wpg;                                          // p.wpg = workgroups per (batch, direction) here
  const int bk = blk / p.wpg;
  if (bk >= p.batch * p.G) return;                                      // (whole workgroup)
  const int grp = bk % p.G, b = bk / p.G;
  const int ugrp = p.ug < p.G ? (int)((p.u_map >> (4 * grp)) & 15) : grp;
  const bool rev = grp < 32 && ((p.rev_mask >> grp) & 1);

  float* sBC = smem;                                   // [2 buf][B, C][16 states][kTS]
  float* wl = smem + kWgBC + wave * kWgWave;
  float* s_dl = wl;                                    // [CH][kTS] delta'
  float* s_du = wl + CH * kTS;                         // [CH][kTS] delta'*u, then y (in place)

  // ---- recurrence identity: lane = (g, c): states 2g, 2g+1 of channel c; a quad = 4 channels of one state group
  const int g = lane >> 3, c = lane & 7, qi = lane & 3;
  const int ch0 = (cwb * nwv + wave) * CH;             // first channel (inside the group) of this wave
  const int hc = ch0 + c;
  const bool cvalid = hc < p.H;
  const bool wactive = ch0 < p.H;                      // a wave beyond the last channel only helps staging B / C
  const int d = grp * p.H + (cvalid ? hc : 0);
  v2f A01, x01 = {0.f, 0.f};
  {
    const float* Ad = p.A + (int64_t)d * kNState + 2 * g;
    A01 = (v2f){Ad[0] * kLog2e, Ad[1] * kLog2e};
  }
  // ---- staging identity of u, delta, out: row r of a 4-row group, float4 column q
  const int r = lane / QL, q = lane % QL;
  const int d0 = grp * p.H + ch0;
  const int d0u = ugrp * p.H + ch0;
  const int nrw = min(CH, p.H - ch0);
  const rsrc_t ru = make_rsrc(p.u + b * p.u_sb + d0u * p.u_sd, ((int64_t)(nrw - 1) * p.u_sd + p.L) * 4);
  const rsrc_t rd = make_rsrc(p.delta + b * p.d_sb + d0 * p.d_sd, ((int64_t)(nrw - 1) * p.d_sd + p.L) * 4);
  const rsrc_t ro = make_rsrc(p.out + ((int64_t)b * p.dim + d0) * p.L, (int64_t)nrw * p.L * 4);
  const rsrc_t rB = make_rsrc(p.B + b * p.B_sb + grp * p.B_sg, ((int64_t)(kNState - 1) * p.B_sn + p.L) * 4);
  const rsrc_t rC = make_rsrc(p.C + b * p.C_sb + grp * p.C_sg, ((int64_t)(kNState - 1) * p.C_sn + p.L) * 4);
  float Dv[NLD], bv[NLD];
  bool rvalid[NLD];
  int uoff[NLD], doff[NLD], ooff[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    rvalid[i] = ch0 + r + RPI * i < p.H;
    const int dd = grp * p.H + (rvalid[i] ? ch0 + r + RPI * i : 0);
    Dv[i] = p.D ? p.D[dd] : 0.f;
    bv[i] = p.bias ? p.bias[dd] : 0.f;
    uoff[i] = (int)((r + RPI * i) * p.u_sd) * 4;
    doff[i] = (int)((r + RPI * i) * p.d_sd) * 4;
    ooff[i] = ((r + RPI * i) * p.L) * 4;
  }
  // ---- staging identity of the shared B / C tile: 2 x 16 rows x 16 quads = 512 quads over the workgroup's threads (>= 256)
  constexpr int NBCQ = 2;
  int bcoff[NBCQ], bclds[NBCQ];
  bool bcok[NBCQ], bcisC[NBCQ];
#pragma unroll
  for (int j = 0; j < NBCQ; ++j) {
    const int k = tid + j * nthreads;
    bcok[j] = k < 512;
    const int isC = (k >> 8) & 1, n = (k >> 4) & 15;
    bcisC[j] = isC;
    bcoff[j] = (int)(n * (isC ? p.C_sn : p.B_sn)) * 4;
    const int qq = k & 15;
    bclds[j] = (isC * kNState + n) * kTS + 4 * (rev ? QL - 1 - qq : qq);
  }
  const int qbc = tid & 15;                            // (k & 15 == tid & 15 for both j: nthreads is a multiple of 64)

  float4 pu[NLD], pd[NLD], pbc[NBCQ];
  auto issue_loads = [&](int t0) {
    const int t = t0 + 4 * q;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      pu[i] = load_quad<true, true>(ru, uoff[i], t, p.L, rev, rvalid[i]);
      pd[i] = load_quad<true, true>(rd, doff[i], t, p.L, rev, rvalid[i]);
    }
#pragma unroll
    for (int j = 0; j < NBCQ; ++j)
      pbc[j] = load_quad<true, true>(bcisC[j] ? rC : rB, bcoff[j], t0 + 4 * qbc, p.L, rev, bcok[j]);
  };
  const int qc = rev ? QL - 1 - q : q;                 // a reversed tile lies mirrored in LDS (quads in memory order)
  float4 yreg[NLD];
  auto store_tile = [&](int t0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) store_quad<true, true>(ro, ooff[i], t0 + 4 * q, p.L, rev, rvalid[i], yreg[i]);
  };
  const int row = lane >> 4;
  const int ystep = (row & 1) * 2 + (row >> 1);        // rows 0..3 end with the sums of steps 0, 2, 1, 3 of a group (see below)

  issue_loads(0);
  for (int tile = 0; tile < p.ntiles; ++tile) {
    const int t0 = tile * kT;
    float* sB = sBC + (tile & 1) * (2 * kNState * kTS);
    // ---- phase 1: registers -> LDS
    float4 uD[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int t = t0 + 4 * q;
      float4 dl, du;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float raw = f4get(pd[i], e) + bv[i];
        float v = SP ? softplus_f(raw) : raw;
        v = (rvalid[i] && t < p.L) ? v : 0.f;
        (&dl.x)[e] = v;
        (&du.x)[e] = v * f4get(pu[i], e);
      }
      uD[i] = make_float4(pu[i].x * Dv[i], pu[i].y * Dv[i], pu[i].z * Dv[i], pu[i].w * Dv[i]);
      const int off = (RPI * i + r) * kTS + 4 * qc;
      *reinterpret_cast<float4*>(s_dl + off) = dl;
      *reinterpret_cast<float4*>(s_du + off) = du;
    }
#pragma unroll
    for (int j = 0; j < NBCQ; ++j)
      if (bcok[j]) *reinterpret_cast<float4*>(sB + bclds[j]) = pbc[j];
    if (tile > 0) store_tile(t0 - kT);
    if (tile + 1 < p.ntiles) issue_loads(t0 + kT);
    // the B / C tile of this step is complete; the other buffer was last read one tile ago, before every wave's previous barrier
    __syncthreads();

    // ---- phase 2: the recurrence
    const int tlen = min(kT, p.L - t0);
    const int ngroups = wactive ? (tlen + 3) >> 2 : 0;
    auto phase2 = [&](auto rvtag) {
      constexpr bool RV = decltype(rvtag)::value;
      struct Ops { float4 dl4, du4; float B0, B1, C0, C1; };
      const float* sBl = sB + (2 * g) * kTS;                   // this lane's two B rows; C rows follow 16 rows later
      auto load_ops = [&](int tg) {
        Ops o;
        const int col = RV ? QL - 1 - tg : tg;
        o.dl4 = *reinterpret_cast<const float4*>(s_dl + c * kTS + 4 * col);
        o.du4 = *reinterpret_cast<const float4*>(s_du + c * kTS + 4 * col);
        const int pos = RV ? kT - 1 - (4 * tg + qi) : 4 * tg + qi;      // lane qi of the quad holds step qi of the group
        o.B0 = sBl[pos]; o.B1 = sBl[kTS + pos];
        o.C0 = sBl[kNState * kTS + pos]; o.C1 = sBl[(kNState + 1) * kTS + pos];
        return o;
      };
      auto at = [](const float4& v, int e) { return f4get(v, RV ? 3 - e : e); };
      auto compute = [&](const Ops& o, int tg) {
        v2f a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const v2f pw = A01 * at(o.dl4, e);
          a[e] = (v2f){__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
        }
        __builtin_amdgcn_sched_barrier(0);
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float du = at(o.du4, e);
          const v2f t = a[e] * x01;
          float tl = t.x, th = t.y;
          fmac_qsel(e, tl, o.B0, du);
          fmac_qsel(e, th, o.B1, du);
          float yy = mul_qsel(e, o.C0, tl);
          fmac_qsel(e, yy, o.C1, th);
          x01 = (v2f){tl, th};
          y[e] = yy;
        }
        // sum over the 8 state groups (lane bits 5, 4, 3): swap32 + add, swap16 + add leave the sums of steps 0, 2, 1, 3 in rows
        // 0..3 (each still split over the two halves of its row), row_ror:8 + add joins the halves
        const auto r01 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, y[0]), __builtin_bit_cast(unsigned, y[1]), false, false);
        const unsigned u0 = r01[0], u1 = r01[1];
        const float s01 = __builtin_bit_cast(float, u0) + __builtin_bit_cast(float, u1);
        const auto r23 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, y[2]), __builtin_bit_cast(unsigned, y[3]), false, false);
        const unsigned u2 = r23[0], u3 = r23[1];
        const float s23 = __builtin_bit_cast(float, u2) + __builtin_bit_cast(float, u3);
        const auto rr = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s01), __builtin_bit_cast(unsigned, s23), false, false);
        const unsigned w0 = rr[0], w1 = rr[1];
        float yt = __builtin_bit_cast(float, w0) + __builtin_bit_cast(float, w1);
        yt += dpp_f<DPP_ROW_ROR8>(yt);
        const int tpos = 4 * tg + ystep;
        s_du[c * kTS + (RV ? kT - 1 - tpos : tpos)] = yt;              // (both halves of a row store the same value)
        if (p.x_chk != nullptr && ((tg & 3) == 3 || tg == ngroups - 1)) {
          // checkpoint (batch, chunk, dim, 16): 8 B per lane, the wave's 8 channels x 16 states are one 512-B run (measured equal
          // to 16-B pieces re-ordered through LDS: 0.369 vs 0.372 ms at the 56x56 stage)
          if (cvalid)
            *reinterpret_cast<float2*>(p.x_chk + (((int64_t)b * p.nchk + (t0 >> 4) + (tg >> 2)) * p.dim + d) * kNState + 2 * g) =
                make_float2(x01.x, x01.y);
        }
      };
      if (ngroups > 0) {
        Ops opA = load_ops(0);
        for (int tg = 0; tg < ngroups; tg += 2) {
          Ops opB = load_ops(min(tg + 1, ngroups - 1));
          __builtin_amdgcn_sched_barrier(0);
          compute(opA, tg);
          __builtin_amdgcn_sched_barrier(0);
          opA = load_ops(min(tg + 2, ngroups - 1));
          __builtin_amdgcn_sched_barrier(0);
          if (tg + 1 < ngroups) compute(opB, tg + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if (rev) phase2(std::true_type{}); else phase2(std::false_type{});

    // ---- phase 3: y (+ D*u) LDS -> registers (stored at the next tile's load point)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const float4 y = *reinterpret_cast<const float4*>(s_du + (RPI * i + r) * kTS + 4 * qc);
      yreg[i] = make_float4(y.x + uD[i].x, y.y + uD[i].y, y.z + uD[i].z, y.w + uD[i].w);
    }
  }
  store_tile((p.ntiles - 1) * kT);
}

template <bool SP>
int launch_wg(const FwdParams& p, int ncw, int nwv, hipStream_t stream) {
  const size_t lds = sizeof(float) * (size_t)(kWgBC + nwv * kWgWave);
  // more than 64 KB of dynamic LDS must be allowed once per kernel and device (idempotent: racing threads set the same value);
  // the largest request is that of 16 waves, so allow it the first time any launch needs the attribute
  static bool allowed[64] = {};
  int dev = 0;
  if (lds > 64 * 1024 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !allowed[dev]) {
    (void)hipFuncSetAttribute((const void*)scan_fwd_wg_kernel<SP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(float) * (size_t)(kWgBC + 16 * kWgWave)));
    allowed[dev] = true;
  }
  FwdParams q = p;
  q.ntiles = (p.L + kWgT - 1) / kWgT;
  q.wpg = ncw;
  const int nblocks = (p.batch * p.G * ncw + 7) & ~7;
  hipLaunchKernelGGL((scan_fwd_wg_kernel<SP>), dim3(nblocks), dim3(nwv * 64), lds, stream, q);
  return (int)hipGetLastError();
}

template <int NS, bool VEC, bool SP, bool LEAN, bool DT = false, bool SH = false>
int launch(const FwdParams& p, int nblocks, int wpb, hipStream_t stream) {
  constexpr int CH = 4 * NS;
  constexpr int TS = fwd_tile(NS, LEAN) + 4;
  constexpr size_t bc = 2 * fwd_tile(NS, LEAN) * bcs_of(NS);                  // one B / C tile: per wave, or two per workgroup (SH)
  const size_t lds = sizeof(float) * ((size_t)wpb * 2 * CH * TS + (SH ? 2 * bc : (size_t)wpb * bc));
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)scan_fwd_kernel<NS, VEC, SP, LEAN, DT, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  FwdParams q = p;
  q.ntiles = (p.L + fwd_tile(NS, LEAN) - 1) / fwd_tile(NS, LEAN);
  hipLaunchKernelGGL((scan_fwd_kernel<NS, VEC, SP, LEAN, DT, SH>), dim3(nblocks), dim3(wpb * 64), lds, stream, q);
  return (int)hipGetLastError();
}

template <int NS, bool LEAN>
int launch_l(const FwdParams& p, int nblocks, int wpb, bool vec, bool sp, bool sh, hipStream_t stream) {
  if (p.dtw != nullptr) return launch<NS, true, true, LEAN, true>(p, nblocks, wpb, stream);   // (checked by the caller: vec && sp)
  if constexpr (NS == 4 && LEAN) {      // workgroup-shared B / C tile (the softplus forms: what SS2D calls)
    if (sh && sp) return vec ? launch<4, true, true, true, false, true>(p, nblocks, wpb, stream)
                             : launch<4, false, true, true, false, true>(p, nblocks, wpb, stream);
  }
  if (vec) return sp ? launch<NS, true, true, LEAN>(p, nblocks, wpb, stream) : launch<NS, true, false, LEAN>(p, nblocks, wpb, stream);
  return sp ? launch<NS, false, true, LEAN>(p, nblocks, wpb, stream) : launch<NS, false, false, LEAN>(p, nblocks, wpb, stream);
}
template <int NS>
int launch_ns(const FwdParams& p, int nblocks, int wpb, bool vec, bool sp, bool lean, bool sh, hipStream_t stream) {
  return lean ? launch_l<NS, true>(p, nblocks, wpb, vec, sp, sh, stream) : launch_l<NS, false>(p, nblocks, wpb, vec, sp, false, stream);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

namespace mm {

// Host-side planning: states per lane.  Fewer states per lane = more wavefronts for narrow problems at the
// price of DPP reduction steps and per-(channel,step) overhead replicated over more lanes.  The VALU reaches
// its 2-cycle issue rate only with >= 2 waves per SIMD (measured), so aim for >= 2048 waves, ideally >= 3072.
int plan_fwd_variant(int batch, int G, int H, int L) {
  (void)L;
  const long seqs = (long)batch * G * H;
  if (seqs / 16 >= 1280) return 4;     // (round 3, packed math: 4 states per lane win from 1536 waves on — S, Bz = 64, 56x56: 0.39 vs 0.50 ms)     // >= 2 waves per SIMD at 4 states per lane (B, Bz = 32, 48x48 stage: 2048 waves of 4
                                       // states 0.33 ms vs 4096 waves of 2 states 0.44 ms); LEAN from 3 per SIMD, see below
  if (seqs / 8 >= 1024) return 2;      // fewer sequences: halve the states per lane to double the wave count
  return 1;
}

int scan_fwd_launch(const mm_scan_args* a, hipStream_t stream, int32_t* plan_out) {
  FwdParams p;
  p.u = a->u; p.delta = a->delta; p.A = a->A; p.B = a->B; p.C = a->C; p.D = a->D; p.bias = a->delta_bias;
  p.out = a->out; p.x_chk = a->x_chk;
  p.u_sb = a->u_sb; p.u_sd = a->u_sd; p.d_sb = a->delta_sb; p.d_sd = a->delta_sd;
  p.B_sb = a->B_sb; p.B_sg = a->B_sg; p.B_sn = a->B_sn; p.C_sb = a->C_sb; p.C_sg = a->C_sg; p.C_sn = a->C_sn;
  p.batch = a->batch; p.dim = a->dim; p.L = a->L; p.G = a->G; p.H = a->dim / a->G;
  p.ntiles = (a->L + kTile - 1) / kTile;
  p.nchk = (a->L + kChunk - 1) / kChunk;
#ifdef MM_EXPERIMENTS
  p.dbg = (a->variant >> 8) & 0xff;
#endif
  const bool shared = a->u_groups > 0 && a->u_groups < a->G;
  p.ug = shared ? a->u_groups : a->G;
  p.u_map = shared ? a->u_map : 0x76543210u;
  p.rev_mask = a->rev_mask;
  int ns = (a->variant & 0xff) ? (a->variant & 0xff) : plan_fwd_variant(a->batch, a->G, p.H, a->L);
  // variant low byte 32: the workgroup-cooperative kernel; it exists for the vector path with softplus and a delta tensor (the
  // training / inference call of SS2D without the fused dt projection) — anything else takes the general kernel with its own plan
  const bool want_wg = ns == 32;
  if (want_wg) ns = plan_fwd_variant(a->batch, a->G, p.H, a->L);
  if (ns != 1 && ns != 2 && ns != 4) return MM_ERR_UNSUPPORTED;
  int wpb = (a->variant >> 16) & 0xff;      // waves per workgroup (tuning knob; waves never synchronise)
  if (wpb <= 0) wpb = (ns == 4 && (long)a->batch * a->dim / 16 >= 3072) ? 4 : 2;
  if (wpb > 4) wpb = 4;
  const int CH = 4 * ns;
  p.wpg = (p.H + CH - 1) / CH;
  // 32-bit byte offsets inside one batch item must not overflow
  const int64_t sdmax = a->u_sd > a->delta_sd ? a->u_sd : a->delta_sd;
  const int64_t span = 16 * (sdmax > a->L ? sdmax : a->L) * 4;       // one wave touches <= 16 rows
  if (span >= 0x7ffffff0ll || (int64_t)kNState * (a->B_sn > a->C_sn ? a->B_sn : a->C_sn) * 4 >= 0x7ffffff0ll)
    return MM_ERR_UNSUPPORTED;
  p.nwaves_total = a->batch * a->G * p.wpg;
  p.dtw = nullptr; p.dts = nullptr; p.dts_sb = p.dts_sg = p.dts_sn = 0; p.R = 0;
  int nblocks = (p.nwaves_total + wpb - 1) / wpb;
  nblocks = (nblocks + 7) & ~7;             // multiple of 8 so the XCD remap is a bijection; surplus waves exit

  const bool vec = (a->L % 4 == 0) && aligned16(a->u) && aligned16(a->delta) && aligned16(a->B) && aligned16(a->C) &&
                   aligned16(a->out) && a->u_sb % 4 == 0 && a->u_sd % 4 == 0 && a->delta_sb % 4 == 0 &&
                   a->delta_sd % 4 == 0 && a->B_sb % 4 == 0 && a->B_sg % 4 == 0 && a->B_sn % 4 == 0 &&
                   a->C_sb % 4 == 0 && a->C_sg % 4 == 0 && a->C_sn % 4 == 0;
  const bool sp = a->delta_softplus != 0;
  if (a->dt_w != nullptr) {     // fused dt projection: vector path with softplus only (the SS2D call: MedMamba.py:262, 273-279)
    if (!a->dts || a->dt_rank <= 0 || a->dt_rank > kDtMax || !vec || !sp || !aligned16(a->dts) || a->dts_sb % 4 || a->dts_sg % 4 ||
        a->dts_sn % 4 || (int64_t)kDtMax * a->dts_sn * 4 >= 0x7ffffff0ll)
      return MM_ERR_UNSUPPORTED;
    p.dtw = a->dt_w; p.dts = a->dts; p.dts_sb = a->dts_sb; p.dts_sg = a->dts_sg; p.dts_sn = a->dts_sn; p.R = a->dt_rank;
  }
  // variant bit 24: force LEAN on, bit 25: force LEAN off; default: lean when the grid offers >= 3 waves per SIMD
  // (measured, S/Bz=64: LEAN wins for NS=4 at >= 3072 waves: 0.19 vs 0.21 ms stage 2, 0.088 vs 0.105 ms stage 3;
  //  it loses for NS=2 on the long stage-1 sequences: 0.49 vs 0.43 ms)
  bool lean = ns == 4 && p.nwaves_total >= 3 * 1024;
  if (a->variant & (1 << 24)) lean = true;
  if (a->variant & (1 << 25)) lean = false;
  // the four waves of a workgroup share one B / C tile when they all belong to one (batch, direction); variant bit 26: private tiles
  const bool sh = lean && ns == 4 && wpb == 4 && p.wpg % 4 == 0 && a->dt_w == nullptr && !(a->variant & (1 << 26));
  // variant low byte 32 (or the default plan, below): the workgroup-cooperative kernel for long sequences with few sequences
  const bool can_wg = vec && sp && a->dt_w == nullptr;
  int wg_ncw = 0, wg_nwv = 0;
  if (can_wg) {
    const int tiles8 = (p.H + 7) / 8;
    wg_ncw = (tiles8 + 15) / 16;
    // fill the chip: at least 256 workgroups if the waves allow it (never below 4 waves per workgroup)
    while ((long)a->batch * a->G * wg_ncw < 256 && (tiles8 + 2 * wg_ncw - 1) / (2 * wg_ncw) >= 4) wg_ncw *= 2;
    wg_nwv = (tiles8 + wg_ncw - 1) / wg_ncw;
    const int wv_req = (a->variant >> 16) & 0xff;   // tuning knob: waves per workgroup
    if (wv_req >= 4 && wv_req <= 16) { wg_nwv = wv_req < tiles8 ? wv_req : tiles8; wg_ncw = (tiles8 + wg_nwv - 1) / wg_nwv; }
    if (wg_nwv < 4) wg_nwv = 4;                     // 512 B / C quads over >= 256 threads
  }
  // default plan: long sequences that offer fewer than 2 wavefronts of 4 states per SIMD (the 56x56 stage of T / S at 64 images:
  // 0.43 -> 0.34 ms inference form, 0.43 -> 0.37 ms with checkpoints; the 96x96 stage of B at 32: 0.91 -> 0.74 / 0.90 -> 0.79 ms);
  // with more sequences the general kernel's 4 states per lane win (28x28 stage: 0.163 vs 0.185 ms).  variant 32 forces it.
  const long seq16 = (long)a->batch * a->G * p.H / 16;
  const bool plan_wg = (a->variant & 0xff) == 0 && a->L >= 512 && seq16 >= 256 && seq16 < 2048;
  const bool use_wg = can_wg && (want_wg || plan_wg);
  if (use_wg) { ns = 2; wpb = wg_nwv; nblocks = (a->batch * a->G * wg_ncw + 7) & ~7; lean = false; }
  if (plan_out) {     // mm_scan_plan: report, do not launch
    plan_out[0] = ns; plan_out[1] = wpb; plan_out[2] = nblocks; plan_out[3] = vec ? 1 : 0; plan_out[4] = lean ? 1 : 0;
    plan_out[5] = use_wg ? 2 : 0;
    return MM_OK;
  }
  if (use_wg) return launch_wg<true>(p, wg_ncw, wg_nwv, stream);
  switch (ns) {
    case 1: return launch_ns<1>(p, nblocks, wpb, vec, sp, lean, false, stream);
    case 2: return launch_ns<2>(p, nblocks, wpb, vec, sp, lean, false, stream);
    default: return launch_ns<4>(p, nblocks, wpb, vec, sp, lean, sh, stream);
  }
}

}  // namespace mm
