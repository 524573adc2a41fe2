// Selective-scan backward for gfx950 (MI355X).  Replaces mamba_ssm's selective_scan_cuda.bwd behind
// SelectiveScanFn.backward (autograd of the call at MedMamba.py:273-279; the adjoint of temp.py:57-139).
//
// Mapping (wave64): workgroup = one (batch, direction) x up to 8 waves; wavefront = 16 channels;
//   lane = (state group g = lane / 16, channel c = lane % 16), 4 of the 16 states per lane.
//   -> sums over the 16 channels of a wave (dB, dC) stay inside a 16-lane DPP row,
//      sums over the 4 state groups (du, ddelta) are two permlane swaps.
// Tiles of 32 steps, last tile first; per 16-step sub-tile (last first): start from the state checkpoint the
// forward kernel wrote (x_chk, every 16 steps), recompute the 16 states into registers, then run the adjoint
// recurrence backwards with 11 VALU + 1 v_exp_f32 per state-step:
//     b = dlu*B;  gxt = C*g + gx;  w = x - b (= a*x_prev);  t2 = gxt*w;  dA += t2*dl;  s1 += t2*A;  s2 += gxt*B;
//     dC_part = g*x;  dB_part = gxt*dlu;  gx = a*gxt              [ddelta' = s1 + u*s2,  du = dl*s2 + D*g]
// dB/dC: an 8-value halving butterfly inside each 16-lane row (22 DPP ops per step instead of 32) leaves one sum
// per lane, which goes to the wave's own LDS slab with a plain store (LDS float atomics are slow: ~40 ns of the CU's
// LDS pipe per ds_add_f32 with 32 active lanes, tools/ubench/lds_rate.cpp); the slabs are double buffered and
// summed over the waves + flushed one tile later, so the whole tile loop has ONE barrier per tile.
// Global traffic uses bounds-checked buffer descriptors (mm_common.h) and a register prefetch of the next
// tile; reversed directions / shared u blocks as in the forward kernel (include/medmamba_hip.h).
#include <type_traits>
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
  int64_t dB_sb, dB_sg, dB_sn, dC_sb, dC_sg, dC_sn;
  int64_t g_sb, o_sb, o_sd;  // batch stride of dout; batch stride of du / ddelta; channel stride shared by all three
  int64_t dpar_sb;          // != 0: dA / dD / dbias are per-batch-item partial buffers (plain stores, no atomics), batch stride
  int64_t dBC_sc;           // != 0 (and ncw > 1): dB / dC go to per-channel-tile partial planes, plane stride
  int dim, L, G, H, CW, ncw, npass, ntiles, nchk;
  int ug;                   // channel blocks in u / dout
  unsigned u_map, rev_mask; // block of group g = (u_map >> 4g) & 15; bit g of rev_mask: group g runs backwards
};

__device__ __forceinline__ float f4get(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

constexpr int T = 32;            // steps per tile
constexpr int TS = T + 4;        // LDS row stride (floats)
constexpr int QL = T / 4;        // float4 columns per row = 8
constexpr int RPI = kWave / QL;  // rows per load instruction = 8
constexpr int NSUB = T / kChunk; // 16-step sub-tiles per tile = 2
// NS = states per lane: 4 (16 channels per wave, 4 state groups) or 2 (8 channels per wave, 8 state groups: twice the waves
// for the same problem at half the registers — the 56x56 stage of T/S and the 96x96 stage of B offer fewer than 2 waves per
// SIMD at 4 states per lane)
constexpr int ch_of(int ns) { return 4 * ns; }                 // channels per wave
constexpr int tsa_of(int ns) { return ns == 4 ? TS : T + 1; }  // row stride of the dB/dC slabs (12 waves of NS = 2 must fit LDS)
constexpr int maxthreads_of(int ns) { return ns == 4 ? 512 : 768; }


// keep + (send of the DPP partner lane)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float keep, float send) { return keep + dpp_f<CTRL>(send); }

template <int NS, bool VEC, bool SP>
__global__ __launch_bounds__(maxthreads_of(NS)) void scan_bwd_kernel(const BwdParams p) {
  constexpr int CH = ch_of(NS);          // channels per wave
  constexpr int NG = kNState / NS;       // state groups = lanes per channel
  constexpr int NLD = CH / RPI;          // row-loads per lane per tensor per tile
  constexpr int TSA = tsa_of(NS);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nthreads = blockDim.x;
  const int cwb = blockIdx.x % p.ncw;
  const int bk = blockIdx.x / p.ncw;
  const int grp = bk % p.G, b = bk / p.G;
  // A workgroup owns channel tiles cwb*npass .. cwb*npass + npass-1 of its (batch, direction) and walks them one after the
  // other (whole sequence each).  With ncw == 1 it owns ALL channels of the direction: dB / dC leave with plain stores in the
  // first pass and plain read-modify-writes afterwards (the same thread touches the same address in every pass) — no
  // atomics, no zero-filled outputs, a fixed summation order.
  for (int pass = 0; pass < p.npass; ++pass) {
  // every per-lane quantity of a pass is derived from this opaque zero, so that the compiler recomputes the pass set-up per
  // pass instead of hoisting it out of the pass loop and keeping it alive through the tile loop (+13 VGPRs = spills at 256)
  int lane_zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
  const int tid = threadIdx.x + lane_zero, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = cwb * p.npass + pass;
  if (cw * p.CW >= p.H && pass > 0) break;          // (workgroup-uniform) nothing left in this direction
  const int ugrp = p.ug < p.G ? (int)((p.u_map >> (4 * grp)) & 15) : grp;
  const bool rev = grp < 32 && ((p.rev_mask >> grp) & 1);

  const int nwaves = nthreads >> 6;
  float* sBC = smem;                                   // [2 buf][2][16][TS]   B, C tiles
  float* sAcc = smem + 2 * 2 * kNState * TS;           // [2 buf][nwaves][2][16][TSA]   dB, dC partial sums, one slab per wave
  float* wl = smem + 2 * 2 * kNState * TS + 2 * nwaves * 2 * kNState * TSA + wave * (3 * CH * TS);
  float* s_u = wl;                                     // u      -> du   (in place)
  float* s_dl = wl + CH * TS;                          // delta'
  float* s_g = wl + 2 * CH * TS;                       // dout -> d(delta') (in place)

  // ---- recurrence identity
  const int g = lane / CH, c = lane % CH;
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
  const float* chk_base = p.x_chk + ((int64_t)b * p.nchk * p.dim + d) * kNState + g * NS;   // (batch, chunk, dim, 16): + ci * dim * 16
  // which of the 2*NS reduced dB/dC sums this lane ends up with (butterfly below).  NS = 4: idx = 4*bit2 + 2*bit3 + bit0 (the
  // lanes with bit1 set hold duplicates and stay out of the store); NS = 2: idx = 2*bit2 + bit1 (duplicates: bit0 set)
  const int ridx = NS == 4 ? ((c >> 2) & 1) * 4 + ((c >> 3) & 1) * 2 + (c & 1) : ((c >> 2) & 1) * 2 + ((c >> 1) & 1);
  const bool acc_writer = NS == 4 ? !(c & 2) : !(c & 1);
  float* acc_lane = sAcc + ((wave * 2 + ridx / NS) * kNState + g * NS + ridx % NS) * TSA;   // + buf*nwaves*2*16*TSA + t

  // ---- staging identity: lane -> (row r of an 8-row group, float4 column q)
  const int r = lane / QL, q = lane % QL;
  const int hc0 = cw * p.CW + wave * CH + r;
  const int d0 = grp * p.H + cw * p.CW + wave * CH;            // first channel of this wave (uniform)
  const int d0u = ugrp * p.H + cw * p.CW + wave * CH;
  // descriptors cover exactly this wave's rows: 32-bit offsets suffice for any channel stride (channel-major planes)
  const int nrw = min(CH, p.H - (cw * p.CW + wave * CH));      // valid rows of this wave (<= 0: empty descriptors)
  const rsrc_t ru = make_rsrc(p.u + b * p.u_sb + d0u * p.u_sd, ((int64_t)(nrw - 1) * p.u_sd + p.L) * 4);
  const rsrc_t rd = make_rsrc(p.delta + b * p.d_sb + d0 * p.d_sd, ((int64_t)(nrw - 1) * p.d_sd + p.L) * 4);
  const rsrc_t rg = make_rsrc(p.dout + b * p.g_sb + d0u * p.o_sd, ((int64_t)(nrw - 1) * p.o_sd + p.L) * 4);
  const rsrc_t rdu = make_rsrc(p.du + b * p.o_sb + d0 * p.o_sd, ((int64_t)(nrw - 1) * p.o_sd + p.L) * 4);
  const rsrc_t rdd = make_rsrc(p.ddelta + b * p.o_sb + d0 * p.o_sd, ((int64_t)(nrw - 1) * p.o_sd + p.L) * 4);
  const rsrc_t rB = make_rsrc(p.B + b * p.B_sb + grp * p.B_sg, ((int64_t)(kNState - 1) * p.B_sn + p.L) * 4);
  const rsrc_t rC = make_rsrc(p.C + b * p.C_sb + grp * p.C_sg, ((int64_t)(kNState - 1) * p.C_sn + p.L) * 4);
  float* dBbase = p.dB + b * p.dB_sb + grp * p.dB_sg + cwb * p.dBC_sc;
  float* dCbase = p.dC + b * p.dC_sb + grp * p.dC_sg + cwb * p.dBC_sc;
  bool rvalid[NLD];
  int uoff[NLD], doff[NLD], ooff[NLD];
  float bv[NLD], dDacc[NLD], dbacc[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    rvalid[i] = hc0 + RPI * i < p.H;
    const int dd = grp * p.H + (rvalid[i] ? hc0 + RPI * i : 0);
    bv[i] = p.bias ? p.bias[dd] : 0.f;
    uoff[i] = (int)((r + RPI * i) * p.u_sd) * 4;
    doff[i] = (int)((r + RPI * i) * p.d_sd) * 4;
    ooff[i] = (int)((r + RPI * i) * p.o_sd) * 4;
    dDacc[i] = 0.f;
    dbacc[i] = 0.f;
  }
  // B/C tile = 2 x 16 rows x 8 quads = 256 float4, spread over the workgroup's threads: one per thread for >= 4 waves, two
  // for 2 or 3 waves (quad index tid, tid + nthreads), both prefetched in registers one tile ahead
  int bc_which[2], bc_n[2], bc_q[2], bc_off[2];
  bool bc_mine[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = tid + k * nthreads;
    bc_which[k] = (idx >> 7) & 1; bc_n[k] = (idx >> 3) & 15; bc_q[k] = idx & 7;
    bc_mine[k] = idx < 256 && (k == 0 || nthreads < 256);
    bc_off[k] = (int)(bc_n[k] * (bc_which[k] ? p.C_sn : p.B_sn)) * 4;
  }
  const bool bc_pref = nthreads >= 128;         // a single wave (tiny problems) stages its four quads per thread without prefetch

  float4 pu[NLD], pd[NLD], pg[NLD], pbc[2];
  auto issue_loads = [&](int t0) {
    const int t = t0 + 4 * q;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      pu[i] = load_quad<VEC, VEC>(ru, uoff[i], t, p.L, rev, rvalid[i]);
      pd[i] = load_quad<VEC, VEC>(rd, doff[i], t, p.L, rev, rvalid[i]);
      pg[i] = load_quad<VEC, VEC>(rg, ooff[i], t, p.L, rev, rvalid[i]);
    }
    if (bc_pref) {
      pbc[0] = load_quad<VEC, VEC>(bc_which[0] ? rC : rB, bc_off[0], t0 + 4 * bc_q[0], p.L, rev, bc_mine[0]);
      if (nthreads < 256) pbc[1] = load_quad<VEC, VEC>(bc_which[1] ? rC : rB, bc_off[1], t0 + 4 * bc_q[1], p.L, rev, bc_mine[1]);
    }
  };
  // Reversed directions on the vector path keep their quads in MEMORY order (no per-component selects on loads / stores);
  // a reversed tile lies mirrored in LDS (time quad q in column QL-1-q, step e in component 3-e) and the loops below
  // address it through colof() / at() — instantiated for both orders, selected by a wave-uniform branch (scan_fwd.hip).
  const bool revm = VEC && rev;
  const int qc = revm ? QL - 1 - q : q;

  // flush one buffer of per-wave partial sums (tile starting at t0) to global: sum over the workgroup's waves.
  // (Plain stores into per-wave slabs + this sum replace LDS float atomics: ds_add_f32 costs ~40 ns of the CU's LDS
  //  pipe per wave-instruction with 32 active lanes — tools/ubench/lds_rate.cpp — i.e. 8 waves x 1 per step = 320 ns of
  //  every ~540 ns step at the 56x56 stage.)
  auto flush_acc = [&](int buf, int t0) {
    // lanes along TIME: a wave-instruction then covers runs of T consecutive dwords of dB / dC rows (128-B segments).  With
    // one float4 (4 steps) per lane the atomics of one instruction were 16 B apart: every 64-B memory-side atomic request
    // carried 4 useful dwords (PMC: the dB/dC atomics cost 4x their bytes in WRITE_SIZE, profiles/r2_scan_traffic_pmc.txt).
    for (int idx = tid; idx < 2 * kNState * T; idx += nthreads) {
      const int which = idx / (kNState * T), n = (idx / T) % kNState, tt = idx % T;
      const float* a = sAcc + ((buf * nwaves * 2 + which) * kNState + n) * TSA + tt;
      float v = a[0];
      for (int w = 1; w < nwaves; ++w) v += a[w * 2 * kNState * TSA];
      const int te = t0 + tt;
      if (te < p.L) {
        float* pdst = (which ? dCbase + n * p.dC_sn : dBbase + n * p.dB_sn) + (rev ? p.L - 1 - te : te);
        // owned rows (one workgroup per direction, or this workgroup's partial plane): a plain store in the first pass, then
        // adds WITHOUT return value — nothing to wait for (a load-add-store here stalls every flush on the load: measured
        // +10-38 % kernel time), and still a fixed order: one thread, one address, program order
        if ((p.ncw == 1 || p.dBC_sc != 0) && pass == 0) *pdst = v; else atomicAdd(pdst, v);
      }
    }
  };

  // (no zero-fill of the slabs: every cell that a flush reads for a time step < L was written by its wave in that tile;
  //  cells of skipped padding groups are never flushed)

  issue_loads((p.ntiles - 1) * T);
  // checkpoint prefetch for the first sub-tile to be processed (the state BEFORE sub-tile `sub` of `tile`)
  auto chk_index = [&](int tile, int sub) { return tile * NSUB + sub - 1; };
  float x0n[NS];
  auto load_chk = [&](int64_t ci, bool ok) {      // NS consecutive states of the checkpoint: one 16-B or 8-B load
    if constexpr (NS == 4) {
      const float4 v = ok ? *reinterpret_cast<const float4*>(chk_base + ci * ((int64_t)p.dim * kNState)) : make_float4(0.f, 0.f, 0.f, 0.f);
      x0n[0] = v.x; x0n[1] = v.y; x0n[2] = v.z; x0n[3] = v.w;
    } else {
      const float2 v = ok ? *reinterpret_cast<const float2*>(chk_base + ci * ((int64_t)p.dim * kNState)) : make_float2(0.f, 0.f);
      x0n[0] = v.x; x0n[1] = v.y;
    }
  };
  {
    const int tl0 = min(T, p.L - (p.ntiles - 1) * T);
    const int ci = chk_index(p.ntiles - 1, (tl0 + kChunk - 1) / kChunk - 1);
    load_chk(ci, ci >= 0 && cvalid);
  }

  for (int tile = p.ntiles - 1; tile >= 0; --tile) {
    const int t0 = tile * T;
    const int buf = (p.ntiles - 1 - tile) & 1;
    // ---- phase 1: registers -> LDS
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int t = t0 + 4 * q;
      float4 dl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float raw = f4get(pd[i], e) + bv[i];
        const bool in = rvalid[i] && (VEC ? t < p.L : t + e < p.L);
        const float v = SP ? softplus_f(raw) : raw;
        (&dl.x)[e] = in ? v : 0.f;
        dDacc[i] = fmaf(f4get(pg[i], e), f4get(pu[i], e), dDacc[i]);
      }
      const int off = (RPI * i + r) * TS + 4 * qc;
      *reinterpret_cast<float4*>(s_u + off) = pu[i];
      *reinterpret_cast<float4*>(s_dl + off) = dl;
      *reinterpret_cast<float4*>(s_g + off) = pg[i];
    }
    if (bc_pref) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (bc_mine[k])
          *reinterpret_cast<float4*>(sBC + ((buf * 2 + bc_which[k]) * kNState + bc_n[k]) * TS + 4 * (revm ? QL - 1 - bc_q[k] : bc_q[k])) = pbc[k];
    } else {   // one wave: every thread stages four quads, no prefetch
      for (int idx = tid; idx < 256; idx += nthreads) {
        const int w_ = (idx >> 7) & 1, n_ = (idx >> 3) & 15, q_ = idx & 7;
        *reinterpret_cast<float4*>(sBC + ((buf * 2 + w_) * kNState + n_) * TS + 4 * (revm ? QL - 1 - q_ : q_)) =
            load_quad<VEC, VEC>(w_ ? rC : rB, (int)(n_ * (w_ ? p.C_sn : p.B_sn)) * 4, t0 + 4 * q_, p.L, rev, true);
      }
    }
    if (tile > 0) issue_loads(t0 - T);
    __syncthreads();                                       // the ONE barrier per tile
    if (tile < p.ntiles - 1) flush_acc(buf ^ 1, t0 + T);   // everybody has finished adding to the previous tile

    // ---- phase 2: sub-tiles of 16 steps, last first
    const int tlen = min(T, p.L - t0);
    const int nsub = (tlen + kChunk - 1) / kChunk;
    const float* sB = sBC + ((buf * 2 + 0) * kNState + g * NS) * TS;
    const float* sC = sBC + ((buf * 2 + 1) * kNState + g * NS) * TS;
    float* accb = acc_lane + buf * nwaves * 2 * kNState * TSA;
    auto phase2 = [&](auto rvtag) {
    constexpr bool RV = decltype(rvtag)::value;
    auto colof = [](int to) { return RV ? T - 4 - to : to; };                  // LDS column (floats) of time offset `to`
    auto at = [](const float4& v, int e) { return f4get(v, RV ? 3 - e : e); };   // time step e of a quad
    for (int sub = nsub - 1; sub >= 0; --sub) {
      const int ts = sub * kChunk;
      // 4-step groups of this sub-tile that hold real time steps (the last sub-tile of a sequence whose length is not a
      // multiple of 16 is partly padding: L = 196 -> 4 of 16 steps, L = 49 -> 1 of 16); the others are skipped entirely
      const int ntq = (min(kChunk, p.L - (t0 + ts)) + 3) >> 2;
      float xs[kChunk][NS];
      float x0[NS];
#pragma unroll
      for (int j = 0; j < NS; ++j) x0[j] = x0n[j];
      {   // prefetch the checkpoint of the NEXT sub-tile to be processed
        int nt = tile, ns_ = sub - 1;
        if (ns_ < 0) { nt = tile - 1; ns_ = NSUB - 1; }
        const int ci = chk_index(nt, ns_);
        load_chk(ci, nt >= 0 && ci >= 0 && cvalid);
      }
      // forward recompute of the 16 states
#pragma unroll
      for (int tq = 0; tq < 4; ++tq) {
        if (tq >= ntq) continue;
        const int to = ts + 4 * tq;
        const float4 dl4 = *reinterpret_cast<const float4*>(s_dl + c * TS + colof(to));
        const float4 u4 = *reinterpret_cast<const float4*>(s_u + c * TS + colof(to));
        float4 Bv[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) Bv[j] = *reinterpret_cast<const float4*>(sB + j * TS + colof(to));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dl = at(dl4, e), dlu = dl * at(u4, e);
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            const float a = __builtin_amdgcn_exp2f(dl * A2[j]);
            const float prev = (tq == 0 && e == 0) ? x0[j] : xs[4 * tq + e - 1][j];
            xs[4 * tq + e][j] = fmaf(a, prev, dlu * at(Bv[j], e));
          }
        }
      }
      // adjoint recurrence, backwards
#pragma unroll
      for (int tq = 3; tq >= 0; --tq) {
        if (tq >= ntq) continue;
        const int to = ts + 4 * tq;
        const float4 dl4 = *reinterpret_cast<const float4*>(s_dl + c * TS + colof(to));
        const float4 u4 = *reinterpret_cast<const float4*>(s_u + c * TS + colof(to));
        const float4 g4 = *reinterpret_cast<const float4*>(s_g + c * TS + colof(to));
        float4 Bv[NS], Cv[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
          Bv[j] = *reinterpret_cast<const float4*>(sB + j * TS + colof(to));
          Cv[j] = *reinterpret_cast<const float4*>(sC + j * TS + colof(to));
        }
        float4 du4, ddl4;
#pragma unroll
        for (int e = 3; e >= 0; --e) {
          const int tt = 4 * tq + e;
          const float dl = at(dl4, e), ut = at(u4, e), gt = at(g4, e);
          const float dlu = dl * ut;
          float s1 = 0.f, s2 = 0.f;
          float v[2 * NS];                             // dB partials [0..NS-1], dC partials [NS..2*NS-1]
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            const float Bn = at(Bv[j], e), Cn = at(Cv[j], e);
            const float xc = xs[tt][j];
            const float gxt = fmaf(Cn, gt, gx[j]);
            gx[j] = __builtin_amdgcn_exp2f(dl * A2[j]) * gxt;   // a_t recomputed: keeping 64 more VGPRs would spill
            // t2 = gxt * a_t * x_{t-1} = gx_new * x_{t-1}: the previous state is in registers, no w = x_t - dlu*B needed — except
            // for the sub-tile's first step, whose predecessor is the checkpoint (not kept: 4 VGPRs short)
            const float t2 = tt == 0 ? gxt * fmaf(-dlu, Bn, xc) : gx[j] * xs[tt == 0 ? 0 : tt - 1][j];
            dAacc[j] = fmaf(t2, dl, dAacc[j]);
            s1 = fmaf(t2, An[j], s1);
            s2 = fmaf(gxt, Bn, s2);
            v[NS + j] = gt * xc;
            v[j] = gxt * dlu;
          }
          // (ddelta', du) partials of this state group -> sum over the 4 rows with two permlane swaps:
          // after swap32+add the lower half holds sum(pa), the upper half sum(pb); swap16 finishes both.
          float pa = fmaf(ut, s2, s1), pb = dl * s2;
          {
            const auto r32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, pa), __builtin_bit_cast(unsigned, pb), false, false);
            const unsigned r0 = r32[0], r1 = r32[1];
            const float s = __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
            const auto r16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s), __builtin_bit_cast(unsigned, s), false, false);
            const unsigned q0 = r16[0], q1 = r16[1];
            pa = __builtin_bit_cast(float, q0) + __builtin_bit_cast(float, q1);   // rows 0,1: ddelta'   rows 2,3: du - D*g
            if constexpr (NS == 2) pa += dpp_f<DPP_ROW_ROR8>(pa);                   // 8 state groups: the two of a 16-lane row
          }
          (&ddl4.x)[RV ? 3 - e : e] = pa;      // d(delta'); the softplus derivative is applied once per element in phase 3
          (&du4.x)[RV ? 3 - e : e] = fmaf(Dc, gt, pa);
          // dB/dC: 2*NS values x CH lanes -> one value per lane (halving butterfly), then one plain store per lane into the slab
          {
            float w1;
            if constexpr (NS == 4) {
            // levels 1 and 2 (8 -> 4 -> 2 values) pair lanes that sit in different DPP banks (bit 2: i <-> i^7 by
            // row_half_mirror; bit 3: i <-> i^8 by row_ror:8), so "keep one half, add the partner's copy of it" is ONE
            // bank-masked v_add_f32_dpp per kept value and bank set — no v_cndmask.  Hand-written: the compiler cannot
            // express a masked add whose untouched lanes keep a third register.  s_nop 1 = the 2 wait states a DPP read
            // needs after a VALU write of the same VGPR (the hazard recogniser does not look inside inline asm); inside
            // the block every DPP source was written >= 3 instructions earlier.
            asm("s_nop 1\n\t"
                "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %2, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %3, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                "v_add_f32_dpp %1, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc"
                : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])
                : "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
            const bool b0 = c & 1;
            w1 = dpp_add<DPP_QUAD_XOR1>(b0 ? v[1] : v[0], b0 ? v[0] : v[1]);   // pair i <-> i^1; bit0 decides
            w1 += dpp_f<DPP_QUAD_XOR2>(w1);           // pair i <-> i^2 (both lanes end with the full sum)
            } else {
              // 4 values x 8 lanes.  Level 1 pairs c <-> 7-c (row_half_mirror inside each 8-lane half row); the lanes c < 4
              // sit in DPP banks 0 / 2, c >= 4 in banks 1 / 3, so one bank-masked add per kept value: c < 4 keeps dB, c >= 4 dC
              asm("s_nop 1\n\t"
                  "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                  "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                  "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                  "v_add_f32_dpp %1, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa"
                  : "+v"(v[0]), "+v"(v[1])
                  : "v"(v[2]), "v"(v[3]));
              const bool b1 = c & 2;
              w1 = dpp_add<DPP_QUAD_XOR2>(b1 ? v[1] : v[0], b1 ? v[0] : v[1]);   // pair i <-> i^2; bit1 decides the state
              w1 += dpp_f<DPP_QUAD_XOR1>(w1);          // pair i <-> i^1 (both lanes end with the full sum)
            }
            if (acc_writer) accb[to + e] = w1;     // this wave's own slab: a plain ds_write_b32
          }
        }
        if (g == 0) *reinterpret_cast<float4*>(s_g + c * TS + colof(to)) = ddl4;   // over the consumed dout quad; delta' stays
        if (g == NG / 2) *reinterpret_cast<float4*>(s_u + c * TS + colof(to)) = du4;
      }
    }
    };   // phase2
    if constexpr (VEC) {
      if (revm) phase2(std::true_type{}); else phase2(std::false_type{});
    } else {
      phase2(std::false_type{});
    }

    // ---- phase 3: private LDS -> global
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int t = t0 + 4 * q;
      const int off = (RPI * i + r) * TS + 4 * qc;
      const float4 vdu = *reinterpret_cast<const float4*>(s_u + off);
      float4 vdd = *reinterpret_cast<const float4*>(s_g + off);
      if (SP) {   // d softplus / d raw = sigmoid(raw) = 1 - exp(-delta')  (series for tiny delta' keeps it relative-accurate)
        const float4 dlq = *reinterpret_cast<const float4*>(s_dl + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dl = f4get(dlq, e);
          (&vdd.x)[e] *= dl < 9.765625e-4f ? dl * (1.f - 0.5f * dl) : 1.f - __builtin_amdgcn_exp2f(-dl * kLog2e);
        }
      }
      store_quad<VEC, VEC>(rdu, ooff[i], t, p.L, rev, rvalid[i], vdu);
      store_quad<VEC, VEC>(rdd, ooff[i], t, p.L, rev, rvalid[i], vdd);
#pragma unroll
      for (int e = 0; e < 4; ++e) dbacc[i] += (rvalid[i] && (VEC ? t < p.L : t + e < p.L)) ? f4get(vdd, e) : 0.f;
    }
  }
  __syncthreads();
  flush_acc((p.ntiles - 1) & 1, 0);               // accumulators of the last processed tile (tile 0)

  // ---- per-channel parameter gradients: reduce in the wave, one atomic per value
  const bool part = p.dpar_sb != 0;     // per-batch-item partial buffers: every (b, d, n) has exactly one writer
  if (cvalid) {
    float* dst = p.dA + (part ? b * p.dpar_sb : 0) + (int64_t)d * kNState + g * NS;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      if (part) dst[j] = dAacc[j]; else atomicAdd(dst + j, dAacc[j]);
    }
  }
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    // the 8 lanes (q) that share a row are 8 consecutive lanes: xor 1, xor 2 and i <-> 7-i cover them
    float sD = dDacc[i], sb = dbacc[i];
    sD += dpp_f<DPP_QUAD_XOR1>(sD); sD += dpp_f<DPP_QUAD_XOR2>(sD); sD += dpp_f<DPP_ROW_HALF_MIRROR>(sD);
    sb += dpp_f<DPP_QUAD_XOR1>(sb); sb += dpp_f<DPP_QUAD_XOR2>(sb); sb += dpp_f<DPP_ROW_HALF_MIRROR>(sb);
    if (q == 0 && rvalid[i]) {
      const int dd = grp * p.H + hc0 + RPI * i;
      if (part) {
        if (p.dD) p.dD[b * p.dpar_sb + dd] = sD;
        if (p.dbias) p.dbias[b * p.dpar_sb + dd] = sb;
      } else {
        if (p.dD) atomicAdd(p.dD + dd, sD);
        if (p.dbias) atomicAdd(p.dbias + dd, sb);
      }
    }
  }
  __syncthreads();      // the next pass reuses the LDS tiles
  }   // pass
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int NS, bool VEC, bool SP>
int launch(const BwdParams& p, int nblocks, int waves, hipStream_t stream) {
  const size_t lds = sizeof(float) * (2 * 2 * kNState * TS + 2 * (size_t)waves * 2 * kNState * tsa_of(NS) + (size_t)waves * 3 * ch_of(NS) * TS);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)scan_bwd_kernel<NS, VEC, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((scan_bwd_kernel<NS, VEC, SP>), dim3(nblocks), dim3(waves * 64), lds, stream, p);
  return (int)hipGetLastError();
}

template <int NS>
int launch_ns(const BwdParams& p, int nblocks, int waves, bool vec, bool sp, hipStream_t stream) {
  if (vec) return sp ? launch<NS, true, true>(p, nblocks, waves, stream) : launch<NS, true, false>(p, nblocks, waves, stream);
  return sp ? launch<NS, false, true>(p, nblocks, waves, stream) : launch<NS, false, false>(p, nblocks, waves, stream);
}
}  // namespace

namespace mm {

int scan_bwd_launch(const mm_scan_args* a, hipStream_t stream, int32_t* plan_out) {
  BwdParams p;
  p.u = a->u; p.delta = a->delta; p.A = a->A; p.B = a->B; p.C = a->C; p.D = a->D; p.bias = a->delta_bias;
  p.x_chk = a->x_chk; p.dout = a->dout; p.du = a->du; p.ddelta = a->ddelta; p.dA = a->dA; p.dB = a->dB;
  p.dC = a->dC; p.dD = a->dD; p.dbias = a->ddelta_bias;
  p.u_sb = a->u_sb; p.u_sd = a->u_sd; p.d_sb = a->delta_sb; p.d_sd = a->delta_sd;
  p.B_sb = a->B_sb; p.B_sg = a->B_sg; p.B_sn = a->B_sn; p.C_sb = a->C_sb; p.C_sg = a->C_sg; p.C_sn = a->C_sn;
  p.dim = a->dim; p.L = a->L; p.G = a->G; p.H = a->dim / a->G;
  const bool dbc_strided = a->dB_sb || a->dB_sg || a->dB_sn || a->dC_sb || a->dC_sg || a->dC_sn;
  p.dB_sn = dbc_strided ? a->dB_sn : a->L;  p.dB_sg = dbc_strided ? a->dB_sg : (int64_t)kNState * a->L;
  p.dB_sb = dbc_strided ? a->dB_sb : (int64_t)a->G * kNState * a->L;
  p.dC_sn = dbc_strided ? a->dC_sn : a->L;  p.dC_sg = dbc_strided ? a->dC_sg : (int64_t)kNState * a->L;
  p.dC_sb = dbc_strided ? a->dC_sb : (int64_t)a->G * kNState * a->L;
  p.ntiles = (a->L + T - 1) / T;
  p.nchk = (a->L + kChunk - 1) / kChunk;
  const bool shared = a->u_groups > 0 && a->u_groups < a->G;
  p.ug = shared ? a->u_groups : a->G;
  p.u_map = shared ? a->u_map : 0x76543210u;
  p.rev_mask = a->rev_mask;
  const bool o_strided = a->dout_sb || a->dud_sb || a->o_sd;
  p.o_sd = o_strided ? a->o_sd : a->L;
  p.g_sb = o_strided ? a->dout_sb : (int64_t)p.ug * p.H * a->L;
  p.o_sb = o_strided ? a->dud_sb : (int64_t)a->dim * a->L;
  int64_t sdmax = a->u_sd > a->delta_sd ? a->u_sd : a->delta_sd;
  if (p.o_sd > sdmax) sdmax = p.o_sd;
  const int64_t span = 16 * (sdmax > a->L ? sdmax : a->L) * 4;       // one wave touches <= 16 rows
  int64_t snmax = a->B_sn > a->C_sn ? a->B_sn : a->C_sn;
  if (p.dB_sn > snmax) snmax = p.dB_sn;
  if (p.dC_sn > snmax) snmax = p.dC_sn;
  if (span >= 0x7ffffff0ll || (int64_t)kNState * snmax * 4 >= 0x7ffffff0ll) return MM_ERR_UNSUPPORTED;
  // states per lane: 2 when 4 would leave the chip below 2 waves per SIMD (the 56x56 stage of T/S at <= 64 images, the 96x96
  // stage of B at 32) — half-width waves, twice as many, 3 per SIMD by their registers.  variant bit 24 / 25 force 2 / 4.
  // Measured (tools/bench_scan.py, ms, 2 vs 4 states per lane): S Bz=64 56x56 1.25 / 1.42, S Bz=32 56x56 0.93 / 1.07, B Bz=32
  // 96x96 2.72 / 2.98; directions wider than 128 channels need three or more workgroups each way and lose (S Bz=32 28x28:
  // 0.48 / 0.38).
  const long waves4 = (long)a->batch * a->G * ((p.H + 15) / 16);
  int ns = (waves4 < 2048 && p.H <= 128) ? 2 : 4;
  if (a->variant & (1 << 24)) ns = 2;
  if (a->variant & (1 << 25)) ns = 4;
  const int CH = ch_of(ns);
  const int waves_needed = (p.H + CH - 1) / CH;
  // waves per workgroup (<= 8 at 4 states per lane, <= 12 at 2: register budget).  A direction that fits ONE workgroup gets one
  // when there are enough (batch, direction) pairs for every CU: dB/dC then leave with plain stores (measured, S/Bz=64 stage
  // 1: 1.51 ms with 256 x 6 waves vs 1.58-1.65 ms with 512 x 3 waves + atomics).  Otherwise 4-wave workgroups — the price is
  // fp32 atomics on dB/dC from the workgroups that share a direction (measured: 4-wave workgroups beat 6- and 8-wave ones at
  // stages 2-4: 0.62 vs 0.76 / 0.80 ms, 0.32 vs 0.34 ms)
  const int wmax = ns == 4 ? 8 : 12;
  const long pairs = (long)a->batch * a->G;
  const int forced = (a->variant >> 16) & 0xff;              // tuning override: waves per workgroup
  const int forced_pass = (a->variant >> 8) & 0xff;          // tuning override: passes per workgroup
  int waves, ncw, npass = 1;
  if (forced == 0 && waves_needed <= wmax && pairs >= 256) {
    waves = waves_needed; ncw = 1;                            // the whole direction in one workgroup, one pass
  } else {
    // 4-wave workgroups (8 at 2 states per lane).  Never fewer: 2-wave workgroups lost everywhere they were measured (B, Bz = 32,
    // 96x96 stage at 4 states per lane: 4.87 ms with 512 x 2 waves, 3.24 ms with 256 x 4, 4.25 ms with 128 x 8); never more
    // when a direction spans several: at the ONE barrier per tile the whole workgroup waits for its slowest wave, and two
    // 4-wave workgroups per CU cover each other's waits (measured at stages 2-4: 0.62 vs 0.76 / 0.80 ms for 6 / 8 waves).
    const int wpw = forced > 0 ? (forced > wmax ? wmax : forced) : (ns == 4 ? 4 : 8);
    const int tiles = (waves_needed + wpw - 1) / wpw;         // channel tiles of wpw waves per direction
    long want = (512 + pairs - 1) / pairs;                    // workgroups per direction that fill the chip
    if (want < 1) want = 1;
    // One pass by default: measured (tools/bench_scan_bwd.py, S, Bz = 64, kernel alone) 0.280 vs 0.290 ms at the 14x14 stage
    // and 0.177 vs 0.187 ms at 7x7 for 1536 / 3072 single-tile workgroups against 512 workgroups walking 3 / 6 tiles — a pass
    // switch drains the whole workgroup, a finished single-tile workgroup is replaced while its neighbour on the CU keeps
    // computing.  Several passes (variant bits 8-15) remain for callers that want fewer dB / dC partial planes.
    (void)want;
    npass = 1;
    if (forced_pass > 0) { npass = forced_pass < tiles ? forced_pass : tiles; }
    ncw = (tiles + npass - 1) / npass;
    waves = (waves_needed + ncw * npass - 1) / (ncw * npass);
  }
  p.CW = waves * CH;
  p.npass = npass;
  p.ncw = ncw;
  // per-batch-item partial buffers for dA / dD / dbias and per-workgroup partial planes for dB / dC (include/medmamba_hip.h):
  // plain stores only — deterministic, nothing to zero-fill
  p.dpar_sb = a->dpar_sb;
  p.dBC_sc = ncw > 1 ? a->dBC_sc : 0;
  const int nblocks = a->batch * a->G * p.ncw;
  const bool vec = (a->L % 4 == 0) && aligned16(a->u) && aligned16(a->delta) && aligned16(a->B) && aligned16(a->C) &&
                   aligned16(a->dout) && aligned16(a->du) && aligned16(a->ddelta) && a->u_sb % 4 == 0 &&
                   a->u_sd % 4 == 0 && a->delta_sb % 4 == 0 && a->delta_sd % 4 == 0 && a->B_sb % 4 == 0 &&
                   a->B_sg % 4 == 0 && a->B_sn % 4 == 0 && a->C_sb % 4 == 0 && a->C_sg % 4 == 0 && a->C_sn % 4 == 0 &&
                   p.o_sd % 4 == 0 && p.g_sb % 4 == 0 && p.o_sb % 4 == 0;
  const bool sp = a->delta_softplus != 0;
  if (plan_out) {     // mm_scan_plan: report, do not launch
    plan_out[0] = ns; plan_out[1] = waves; plan_out[2] = nblocks; plan_out[3] = vec ? 1 : 0; plan_out[4] = 0;
    plan_out[5] = (ncw == 1 || a->dBC_sc != 0) && a->dpar_sb != 0;      // no atomics anywhere: bitwise reproducible
    plan_out[6] = ncw; plan_out[7] = npass;
    return MM_OK;
  }
  return ns == 4 ? launch_ns<4>(p, nblocks, waves, vec, sp, stream) : launch_ns<2>(p, nblocks, waves, vec, sp, stream);
}

}  // namespace mm
