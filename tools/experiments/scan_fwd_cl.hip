// Selective-scan forward for gfx950, "channel-lane" mapping (second generation; scan_fwd.hip holds the first).
// Same operator (selective_scan_fn, MedMamba.py:273-279; arithmetic temp.py:57-139), same C ABI (mm_scan_fwd).
//
// Why a second mapping: B_t[n] and C_t[n] are shared by every channel of a direction (temp.py:95-98), so with
// lane = channel they are WAVE-UNIFORM and can be SGPR operands of the VALU instructions: the recurrence then costs
// exactly v_mul, v_exp_f32, v_mul, v_fma, v_fma per state-step — no per-lane LDS reads of B/C and no DPP reductions
// (the first-generation kernel spends 7-9 VALU per state-step on them, profiles/r1_scan_fwd_pmc_v2.txt).
//
//   workgroup = one (batch, direction) x 64 consecutive channels, W = 16/NS wavefronts (NS = 16: ONE wavefront, no barrier)
//   wavefront g = state group g: states g*NS .. g*NS+NS-1 of ALL 64 channels, in VGPRs for the whole sequence
//   lane      = channel
// Per tile of CT steps (32; 16 for the single-wave form):
//   staging   (all waves): u, delta rows by buffer_load_dwordx4 -> softplus, delta*u -> LDS [channel][t] (row stride
//             CT+4 floats: conflict-free b128 rows)                                                  | barrier
//   recurrence (wave g, 4 steps per group): delta', delta'*u of the lane's channel by 2 ds_read_b128; B/C of the wave's
//             states by s_buffer_load_dwordx4 (scalar cache -> SGPRs, one quad = 4 steps of one state; at most 8 states =
//             64 SGPRs at a time: NS = 16 makes two passes per group), next group's operands in flight while the current
//             group computes where the SGPR budget allows (NS <= 4); the wave's partial y quad = sum_{n in group} C x is
//             written to the wave's OWN LDS slab [channel][t] (ds_write_b128; NS = 16: in place of delta'*u).  LDS float
//             atomics are not an option: measured 170 cycles per ds_add_f32 wave-instruction per CU (tools/ubench/sgpr_rate.cpp) | barrier
//   epilogue  (all waves): sum of the W slabs (+ D*u kept in registers) -> buffer_store_dwordx4 with the staging mapping.
// Reversed directions / shared u blocks (cross-scan without materialisation) as in scan_fwd.hip: time step t is memory
// position L-1-t; the B/C quads of a reversed direction are consumed component 3..0 (compile-time: the body is
// instantiated for both orders and selected by a wave-uniform branch).
// Tails: buffer descriptors range-check every vector access (reads 0 / drops stores); padded steps are identity steps
// (delta' = 0 -> a = 1, b = 0); scalar buffer loads are range-checked by the same kind of descriptor.
#include <type_traits>
#include "mm_common.h"
#include "medmamba_hip.h"

namespace {
using namespace mm;

typedef int v4i __attribute__((ext_vector_type(4)));

struct ClParams {
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
  int batch, dim, L, G, H;
  int ctiles;               // 64-channel tiles per (batch, group)
  int ug;
  unsigned u_map, rev_mask;
  int ntiles, nchk;
  int nwg_total;
  int dbg;
};

constexpr int cl_tile(int ns) { return ns >= 16 ? 16 : 32; }   // steps per tile (the single-wave form stages its 64 rows alone)

__device__ __forceinline__ float f4g(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// scalar (SMEM) buffer loads: wave-uniform descriptor + byte offset -> 4 SGPRs.  The compiler does not track them:
// every consumer is made to depend on sbuf_wait() below.
// (the descriptor is the same __amdgpu_buffer_rsrc_t the vector buffer loads use: it is an SGPR quad either way)
__device__ __forceinline__ v4f sbuf_load4(rsrc_t rsrc, int byte_off) {
  v4f d;
  asm volatile("s_buffer_load_dwordx4 %0, %1, %2" : "=s"(d) : "s"(rsrc), "s"(byte_off));
  return d;
}
__device__ __forceinline__ float sbuf_load1(rsrc_t rsrc, int byte_off) {
  float d;
  asm volatile("s_buffer_load_dword %0, %1, %2" : "=s"(d) : "s"(rsrc), "s"(byte_off));
  return d;
}

// Operands of one pass (HS <= 8 states) of one 4-step group of one wavefront: B/C quads in SGPRs, delta' / delta'*u of
// the lane's channel in VGPRs.  All of it is loaded by inline asm (s_buffer_load / ds_read_b128) so that ONE hand-placed
// s_waitcnt lgkmcnt(0) per pass covers everything: scalar loads return out of order, so a counted wait is not
// possible while one is in flight, and compiler-visible LDS reads next to them would be waited for with a count
// that drains the scalar prefetch as well.
template <int HS>
struct GroupOps { v4f b[HS], c[HS]; };

__device__ __forceinline__ v4f lds_read4(unsigned byte_addr) {
  v4f d;
  asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(byte_addr));
  return d;
}

// (an asm with any VGPR output makes ALL its outputs divergent for the compiler — the SGPR quads would be copied to
// VGPRs — hence separate statements; volatile asm statements keep their order)
template <int HS>
__device__ __forceinline__ void ops_wait(GroupOps<HS>& q) {
  if constexpr (HS == 4)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+s"(q.b[0]), "+s"(q.b[1]), "+s"(q.b[2]), "+s"(q.b[3]), "+s"(q.c[0]), "+s"(q.c[1]), "+s"(q.c[2]), "+s"(q.c[3]));
  else
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+s"(q.b[0]), "+s"(q.b[1]), "+s"(q.b[2]), "+s"(q.b[3]), "+s"(q.b[4]), "+s"(q.b[5]), "+s"(q.b[6]), "+s"(q.b[7]),
                   "+s"(q.c[0]), "+s"(q.c[1]), "+s"(q.c[2]), "+s"(q.c[3]), "+s"(q.c[4]), "+s"(q.c[5]), "+s"(q.c[6]), "+s"(q.c[7]));
}
__device__ __forceinline__ void vec_wait(v4f& a, v4f& b) { asm volatile("" : "+v"(a), "+v"(b)); }

// NS: states per wavefront (4, 8, 16; W = 16/NS wavefronts per workgroup).  VEC: L % 4 == 0 and 16-B aligned rows.  SP: softplus.
template <int NS, bool VEC, bool SP>
__global__ __launch_bounds__(64 * (kNState / NS)) void scan_fwd_cl_kernel(const ClParams p) {
  constexpr int W = kNState / NS;
  constexpr int NT = 64 * W;
  constexpr int CT = cl_tile(NS);                // steps per tile
  constexpr int CTS = CT + 4;                    // LDS row stride of the [channel][t] tiles
  constexpr int CQL = CT / 4;                    // float4 columns (lanes) per row
  constexpr int NQ = (64 * CQL) / NT;            // staged float4 per thread per tensor per tile
  constexpr int RPP = NT / CQL;                  // rows per staging pass
  constexpr int HS = NS > 8 ? 8 : NS;            // states per SGPR pass
  constexpr int NP = NS / HS;                    // passes per 4-step group
  constexpr bool DB = HS <= 4;                   // two SGPR operand sets (next group's B/C in flight during the arithmetic)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_dl = smem;                            // [64][CTS] delta'
  float* s_du = smem + 64 * CTS;                 // [64][CTS] delta' * u   (W == 1: overwritten in place by y)
  float* s_y = smem + 2 * 64 * CTS;              // [W][64][CTS] partial y of every state group (W > 1)

  const int tid = threadIdx.x, lane = tid & 63;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 6);       // state group of this wavefront
  int blk = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) blk = (blk & 7) * (nblk >> 3) + (blk >> 3);   // the workgroups of one (batch, direction) share an XCD's L2
  if (blk >= p.nwg_total) return;
  const int ct = blk % p.ctiles;
  const int bk = blk / p.ctiles;
  const int grp = bk % p.G, b = bk / p.G;
  const int ugrp = p.ug < p.G ? (int)((p.u_map >> (4 * grp)) & 15) : grp;
  const bool rev = grp < 32 && ((p.rev_mask >> grp) & 1);

  // ---- recurrence identity: lane = channel
  const int hc = ct * 64 + lane;
  const bool cvalid = hc < p.H;
  const int d = grp * p.H + (cvalid ? hc : 0);
  float A2[NS], x[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    A2[j] = p.A[(int64_t)d * kNState + g * NS + j] * kLog2e;
    x[j] = 0.f;
  }
  // B/C rows of this wave's states: scalar descriptors over the 16 rows of this (batch, direction)
  const rsrc_t sB = make_rsrc(p.B + b * p.B_sb + grp * p.B_sg, ((int64_t)(kNState - 1) * p.B_sn + p.L) * 4);
  const rsrc_t sC = make_rsrc(p.C + b * p.C_sb + grp * p.C_sg, ((int64_t)(kNState - 1) * p.C_sn + p.L) * 4);
  const int bRow = (int)((g * NS) * p.B_sn) * 4, cRow = (int)((g * NS) * p.C_sn) * 4;   // byte offset of state g*NS
  const int bStep = (int)p.B_sn * 4, cStep = (int)p.C_sn * 4;

  // ---- staging identity: thread -> NQ (row, float4 column) slots
  const int sr = tid / CQL, sq = tid % CQL;      // row within a pass, float4 column
  const int d0 = grp * p.H + ct * 64, d0u = ugrp * p.H + ct * 64;
  const int nrw = min(64, p.H - ct * 64);
  const rsrc_t ru = make_rsrc(p.u + b * p.u_sb + d0u * p.u_sd, ((int64_t)(nrw - 1) * p.u_sd + p.L) * 4);
  const rsrc_t rd = make_rsrc(p.delta + b * p.d_sb + d0 * p.d_sd, ((int64_t)(nrw - 1) * p.d_sd + p.L) * 4);
  const rsrc_t ro = make_rsrc(p.out + ((int64_t)b * p.dim + d0) * p.L, (int64_t)nrw * p.L * 4);
  float Dv[NQ], bv[NQ];
  bool rvalid[NQ];
  int uoff[NQ], doff[NQ], ooff[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int row = sr + RPP * i;
    rvalid[i] = ct * 64 + row < p.H;
    const int dd = grp * p.H + (rvalid[i] ? ct * 64 + row : 0);
    Dv[i] = p.D ? p.D[dd] : 0.f;
    bv[i] = p.bias ? p.bias[dd] : 0.f;
    uoff[i] = (int)(row * p.u_sd) * 4;
    doff[i] = (int)(row * p.d_sd) * 4;
    ooff[i] = (row * p.L) * 4;
  }
  // B/C reach the SGPRs through the scalar cache; a line that is not even in L2 costs a scalar load ~1 us (HBM miss),
  // which no SGPR-budget-sized prefetch can cover.  So the lines of the tile AFTER the next one are pulled into L2 by
  // vector loads whose data nobody uses (one dword per 64-B line: 32 rows x CT*4/64 lines, first threads only).
  constexpr int LPR = (CT * 4 + 63) / 64 + 1;     // 64-B lines per row per tile (+1: rows are not line-aligned)
  const bool toucher = tid < 2 * kNState * LPR && !(p.dbg & 16);
  const int trow = (tid / LPR) % kNState, tline = tid % LPR;
  const bool tC = tid >= kNState * LPR;
  const int toff = (int)(trow * (tC ? p.C_sn : p.B_sn)) * 4 + tline * 64;
  unsigned touch = 0;
  float4 pu[NQ], pd[NQ];
  auto issue_loads = [&](int t0) {
    {   // L2 warm-up of the B/C lines of the tile that starts at t0 + CT (older than the loads below: retired with them)
      const int tt = t0 + CT;
      const int pos = rev ? p.L - tt - CT : tt;
      touch = __builtin_amdgcn_raw_buffer_load_b32(tC ? sC : sB, (toucher && tt < p.L) ? toff + max(pos, 0) * 4 : kOOB, 0, 0);
    }
    const int t = t0 + 4 * sq;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      pu[i] = load_quad<VEC>(ru, uoff[i], t, p.L, rev, rvalid[i]);
      pd[i] = load_quad<VEC>(rd, doff[i], t, p.L, rev, rvalid[i]);
    }
  };

  float* chk = p.x_chk ? p.x_chk + ((int64_t)b * p.dim + d) * p.nchk * kNState + g * NS : nullptr;
  const unsigned lds_dl = (unsigned)reinterpret_cast<uintptr_t>(s_dl + lane * CTS);
  const unsigned lds_du = (unsigned)reinterpret_cast<uintptr_t>(s_du + lane * CTS);
  float* my_y = (W > 1 ? s_y + g * 64 * CTS : s_du) + lane * CTS;      // this lane's row of this wave's y slab

  // issue the B/C loads of states [s0, s0 + HS) of the 4-step group that starts at time step t
  auto issue_bc = [&](GroupOps<HS>& q, int s0, int t, bool rv) {
    if (VEC || !rv || t + 4 <= p.L) {
      const int pos = (p.dbg & 4) ? 0 : (rv ? p.L - 4 - t : t);      // dbg 4: timing-only ablation, every scalar load hits its cache
#pragma unroll
      for (int j = 0; j < HS; ++j) {
        q.b[j] = sbuf_load4(sB, bRow + (s0 + j) * bStep + pos * 4);
        q.c[j] = sbuf_load4(sC, cRow + (s0 + j) * cStep + pos * 4);
      }
    } else {   // reversed direction, L % 4 != 0, last (partial) group: positions below 0 must not be touched
#pragma unroll
      for (int j = 0; j < HS; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pos = p.L - 1 - (t + e);                  // component 3-e of the quad in memory order
          const int ob = pos >= 0 ? bRow + (s0 + j) * bStep + pos * 4 : 0x7ffffff0;
          const int oc = pos >= 0 ? cRow + (s0 + j) * cStep + pos * 4 : 0x7ffffff0;
          q.b[j][3 - e] = sbuf_load1(sB, ob);
          q.c[j][3 - e] = sbuf_load1(sC, oc);
        }
      }
    }
  };

  // checkpoint of this wave's NS states (x_chk rows are 64-B aligned, so the NS-float slice is naturally aligned)
  auto store_state = [&](float* dst) {
#pragma unroll
    for (int j = 0; j < NS; j += 4) *reinterpret_cast<float4*>(dst + j) = make_float4(x[j], x[j + 1], x[j + 2], x[j + 3]);
  };

  // one pass (states s0 .. s0+HS-1) over one 4-step group.  REV selects the component order of the SGPR quads.
  auto pass = [&](auto revtag, auto s0tag, const GroupOps<HS>& q, const v4f dl, const v4f du4, v4f& y4) {
    constexpr bool REV = decltype(revtag)::value;
    constexpr int S0 = decltype(s0tag)::value;
    float a[4][HS];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int j = 0; j < HS; ++j) a[e][j] = __builtin_amdgcn_exp2f(dl[e] * A2[S0 + j]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float du = du4[e];
      float y = y4[e];
#pragma unroll
      for (int j = 0; j < HS; ++j) {
        const float Bn = q.b[j][REV ? 3 - e : e], Cn = q.c[j][REV ? 3 - e : e];
        x[S0 + j] = fmaf(a[e][j], x[S0 + j], du * Bn);
        y = fmaf(x[S0 + j], Cn, y);
      }
      y4[e] = y;
    }
  };

  auto tile_body = [&](auto revtag, int t0, int ngroups) {
    constexpr bool REV = decltype(revtag)::value;
    if (ngroups <= 0) return;
    auto after_group = [&](int tg, v4f y4) {
      if (!(p.dbg & 8)) *reinterpret_cast<v4f*>(my_y + 4 * tg) = y4;   // ds_write_b128 into the wave's own slab row
      if (chk != nullptr && (tg & 3) == 3 && cvalid) store_state(chk + (int64_t)((t0 >> 4) + (tg >> 2)) * kNState);
    };
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (DB) {
      GroupOps<HS> qa, qb;
      v4f dla, dua, dlb, dub;
      issue_bc(qa, 0, t0, REV);
      dla = lds_read4(lds_dl); dua = lds_read4(lds_du);
      for (int tg = 0; tg < ngroups; tg += 2) {
        ops_wait(qa); vec_wait(dla, dua);
        // the prefetch is unconditional (past the tile / sequence end it reads the LDS row pad and range-checked zeros):
        // a conditionally executed asm whose SGPR results are carried around the loop does not compile
        // ("illegal VGPR to SGPR copy": the compiler places the phi of the asm results in VGPRs)
        issue_bc(qb, 0, t0 + 4 * (tg + 1), REV);
        dlb = lds_read4(lds_dl + 16 * (tg + 1)); dub = lds_read4(lds_du + 16 * (tg + 1));
        v4f y4 = zero4;
        pass(revtag, std::integral_constant<int, 0>{}, qa, dla, dua, y4);
        after_group(tg, y4);
        if (tg + 1 < ngroups) {
          ops_wait(qb); vec_wait(dlb, dub);
          issue_bc(qa, 0, t0 + 4 * (tg + 2), REV);
          dla = lds_read4(lds_dl + 16 * (tg + 2)); dua = lds_read4(lds_du + 16 * (tg + 2));
          y4 = zero4;
          pass(revtag, std::integral_constant<int, 0>{}, qb, dlb, dub, y4);
          after_group(tg + 1, y4);
        }
      }
      // the last unconditional prefetches are still in flight: their destination registers must stay reserved until
      // they have landed (a scalar load that lands after its SGPRs were handed to something else corrupts that value)
      ops_wait(qa); ops_wait(qb); vec_wait(dla, dua); vec_wait(dlb, dub);
    } else {
      // one SGPR set (64 SGPRs hold 8 states x 4 steps x B,C); the LDS operands of the next group are still prefetched
      v4f dla = lds_read4(lds_dl), dua = lds_read4(lds_du), dlb = dla, dub = dua;
      for (int tg = 0; tg < ngroups; ++tg) {
        GroupOps<HS> q0;
        issue_bc(q0, 0, t0 + 4 * tg, REV);
        ops_wait(q0); vec_wait(dla, dua);
        dlb = lds_read4(lds_dl + 16 * (tg + 1)); dub = lds_read4(lds_du + 16 * (tg + 1));     // (past the end: row pad)
        v4f y4 = zero4;
        pass(revtag, std::integral_constant<int, 0>{}, q0, dla, dua, y4);
        if constexpr (NP == 2) {
          GroupOps<HS> q1;
          issue_bc(q1, HS, t0 + 4 * tg, REV);
          ops_wait(q1); vec_wait(dlb, dub);      // (also retires the LDS prefetch: it had the first pass to land)
          pass(revtag, std::integral_constant<int, HS>{}, q1, dla, dua, y4);
        }
        after_group(tg, y4);
        if constexpr (NP == 1) {       // the LDS prefetch had the whole pass to land; it must have before the copy below
          asm volatile("s_waitcnt lgkmcnt(0)");
          vec_wait(dlb, dub);
        }
        dla = dlb; dua = dub;
      }
      asm volatile("s_waitcnt lgkmcnt(0)");      // the LDS prefetch past the last group
      vec_wait(dla, dua);
    }
    // a sequence that ends inside a 16-step chunk: the state after its last step is that chunk's checkpoint
    if (chk != nullptr && (ngroups & 3) != 0 && cvalid) store_state(chk + (int64_t)((t0 >> 4) + (ngroups >> 2)) * kNState);
  };

  issue_loads(0);
  for (int tile = 0; tile < p.ntiles; ++tile) {
    const int t0 = tile * CT;
    // ---- staging: registers -> LDS (delta', delta'*u); D*u stays in registers for the epilogue
    float4 uD[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int t = t0 + 4 * sq;
      float4 dl, du;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float raw = f4g(pd[i], e) + bv[i];
        float v = SP ? softplus_f(raw) : raw;
        v = (rvalid[i] && t + e < p.L) ? v : 0.f;            // identity step outside the sequence / channel range
        (&dl.x)[e] = v;
        (&du.x)[e] = v * f4g(pu[i], e);
      }
      uD[i] = make_float4(pu[i].x * Dv[i], pu[i].y * Dv[i], pu[i].z * Dv[i], pu[i].w * Dv[i]);
      const int off = (sr + RPP * i) * CTS + 4 * sq;
      *reinterpret_cast<float4*>(s_dl + off) = dl;
      *reinterpret_cast<float4*>(s_du + off) = du;
    }
    asm volatile("" :: "v"(touch));                          // the warm-up load is a real load (its value is not used)
    if (tile + 1 < p.ntiles) issue_loads(t0 + CT);            // in flight during the recurrence
    __syncthreads();      // (W == 1: one wavefront — its LDS accesses are in order, this is only the compiler's fence)

    // ---- recurrence
    const int tlen = min(CT, p.L - t0);
    const int ngroups = (p.dbg & 2) ? 0 : (tlen + 3) >> 2;
    if (rev) tile_body(std::true_type{}, t0, ngroups); else tile_body(std::false_type{}, t0, ngroups);
    __syncthreads();

    // ---- epilogue: sum of the state groups' slabs (+ D*u) -> global with the staging mapping
    const bool st_en = !(p.dbg & 1);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int off = (sr + RPP * i) * CTS + 4 * sq;
      float4 y = uD[i];
      if constexpr (W == 1) {
        const float4 v = *reinterpret_cast<const float4*>(s_du + off);
        y.x += v.x; y.y += v.y; y.z += v.z; y.w += v.w;
      } else {
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const float4 v = *reinterpret_cast<const float4*>(s_y + w * 64 * CTS + off);
          y.x += v.x; y.y += v.y; y.z += v.z; y.w += v.w;
        }
      }
      store_quad<VEC>(ro, ooff[i], t0 + 4 * sq, p.L, rev, rvalid[i] && st_en, y);
    }
    // (the next tile's staging writes s_dl / s_du only; the slabs are rewritten after the next barrier, and a wave reaches
    //  that barrier only after its epilogue reads above)
  }
}

template <int NS, bool VEC, bool SP>
int launch_cl(const ClParams& p, int nblocks, hipStream_t stream) {
  constexpr int W = kNState / NS;
  constexpr int CTS = cl_tile(NS) + 4;
  const size_t lds = sizeof(float) * (2 * 64 * CTS + (W > 1 ? W * 64 * CTS : 0));
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)scan_fwd_cl_kernel<NS, VEC, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  ClParams q = p;
  q.ntiles = (p.L + cl_tile(NS) - 1) / cl_tile(NS);
  hipLaunchKernelGGL((scan_fwd_cl_kernel<NS, VEC, SP>), dim3(nblocks), dim3(64 * W), lds, stream, q);
  return (int)hipGetLastError();
}
template <int NS>
int launch_cl_ns(const ClParams& p, int nblocks, bool vec, bool sp, hipStream_t stream) {
  if (vec) return sp ? launch_cl<NS, true, true>(p, nblocks, stream) : launch_cl<NS, true, false>(p, nblocks, stream);
  return sp ? launch_cl<NS, false, true>(p, nblocks, stream) : launch_cl<NS, false, false>(p, nblocks, stream);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

namespace mm {

// ns: states per wavefront (4, 8, 16)
int scan_fwd_cl_launch(const mm_scan_args* a, int ns, hipStream_t stream) {
  ClParams p;
  p.u = a->u; p.delta = a->delta; p.A = a->A; p.B = a->B; p.C = a->C; p.D = a->D; p.bias = a->delta_bias;
  p.out = a->out; p.x_chk = a->x_chk;
  p.u_sb = a->u_sb; p.u_sd = a->u_sd; p.d_sb = a->delta_sb; p.d_sd = a->delta_sd;
  p.B_sb = a->B_sb; p.B_sg = a->B_sg; p.B_sn = a->B_sn; p.C_sb = a->C_sb; p.C_sg = a->C_sg; p.C_sn = a->C_sn;
  p.batch = a->batch; p.dim = a->dim; p.L = a->L; p.G = a->G; p.H = a->dim / a->G;
  p.ntiles = 0;
  p.nchk = (a->L + kChunk - 1) / kChunk;
  p.dbg = (a->variant >> 8) & 0xff;
  const bool shared = a->u_groups > 0 && a->u_groups < a->G;
  p.ug = shared ? a->u_groups : a->G;
  p.u_map = shared ? a->u_map : 0x76543210u;
  p.rev_mask = a->rev_mask;
  if (ns != 4 && ns != 8) return MM_ERR_UNSUPPORTED;   // (the single-wave NS = 16 form fails the reversed-direction parity test: not offered)
  p.ctiles = (p.H + 63) / 64;
  const int64_t sdmax = a->u_sd > a->delta_sd ? a->u_sd : a->delta_sd;
  const int64_t span = 64 * (sdmax > a->L ? sdmax : a->L) * 4;       // one workgroup touches <= 64 rows
  if (span >= 0x7ffffff0ll || (int64_t)kNState * (a->B_sn > a->C_sn ? a->B_sn : a->C_sn) * 4 >= 0x7ffffff0ll)
    return MM_ERR_UNSUPPORTED;
  p.nwg_total = a->batch * a->G * p.ctiles;
  const int nblocks = (p.nwg_total + 7) & ~7;                          // multiple of 8: the XCD remap stays a bijection
  const bool vec = (a->L % 4 == 0) && aligned16(a->u) && aligned16(a->delta) && aligned16(a->out) && a->u_sb % 4 == 0 &&
                   a->u_sd % 4 == 0 && a->delta_sb % 4 == 0 && a->delta_sd % 4 == 0;
  const bool sp = a->delta_softplus != 0;
  return ns == 4 ? launch_cl_ns<4>(p, nblocks, vec, sp, stream) : launch_cl_ns<8>(p, nblocks, vec, sp, stream);
}

}  // namespace mm
