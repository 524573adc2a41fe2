// EXPERIMENT (round 4, not built into the library): the "rows" form of the forward scan — parity-green (it ran as `variant` 16 of
// mm_scan_fwd through tests/test_scan_parity.py at commit 9396b0c), slower than the general kernel: S, 64 images, ms per call,
// inference / training form: 56x56 0.459 / 0.509 vs 0.452 / 0.428; 28x28 0.200 / 0.230 vs 0.166 / 0.175; 14x14 0.094 / 0.114 vs
// 0.078 / 0.100 (182 VGPRs: 2 waves per SIMD where the register-lean general kernel runs 4; B / C prefetched only one 16-step block
// ahead; checkpoint stores in 16-B pieces 64 B apart).  Its inner loop alone (tools/ubench/fwd_loop.hip, "D3") is as fast as the
// general kernel's (41-42 ns per wave-step of 4 states at 2-3 waves per SIMD).  What came out of it: the workgroup-cooperative kernel
// in csrc/scan_fwd.hip took the DPP-broadcast operand delivery (as quad broadcasts) and the permlane y reduction.
// This file is the code as it stood inside csrc/scan_fwd.hip (same translation unit: FwdParams, load_quad, ... from there).
// ======================================================================================================================
// "rows" form of the same scan (round 4): lane = (state group g = lane / 16 -> a DPP row, channel c = lane % 16), 4 states per
// lane, 16 channels per wavefront, vector path + softplus only (the SS2D call).  What differs from the kernel above:
//   * B and C never touch LDS.  Per 16-step block a lane loads ITS states' B / C at time step (block start + c) straight from
//     global memory / L2 (8 buffer_load_dword, 64-B runs per state, any alignment, reversed directions by address) and the
//     recurrence's FMAs read step s of the block through DPP `row_newbcast:s` (every lane of a 16-lane row reads lane s of its
//     row — exactly "the state group's B at step s"): no staging stores (the old tile took 32 conflicted ds_write_b32 per lane
//     and tile), no ds_read_b128 per step (2 of the 2.5 LDS reads per step and wave were B / C), 8 VGPRs per block instead
//     of 32 per 4 steps.
//   * y is summed over the four rows with three permlane swaps + three adds per 4 steps (12 DPP / select operations before), and
//     every lane stores one of the four sums — no EXEC masking.
//   * the two multiplies of a state pair stay packed (v_pk_mul_f32); the two FMAs per state carry the DPP operand (VOP2 DPP
//     has no packed form).
// LDS per wave: the delta' and delta'*u / y tiles only (8.7 KB for 64 steps): occupancy is set by registers alone.
// Micro-benchmark of the inner loop alone, tools/ubench/fwd_loop.hip.
template <int S> __device__ __forceinline__ void fmac_bc(float& acc, float b, float m) {     // acc += (lane S of b's row) * m
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(S));
}
template <int S> __device__ __forceinline__ float mul_bc(float b, float m) {
  float r;
  asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(b), "v"(m), "n"(S));
  return r;
}
// the step index is a constant after unrolling: the switch folds to one case.  The DPP operand (b) is only ever written by a
// buffer load (never by a VALU instruction), so the 2-wait-state VALU-write -> DPP-read hazard, which the hazard recogniser
// cannot see inside inline asm, does not arise.
__device__ __forceinline__ void fmac_sel(int s, float& acc, float b, float m) {
  switch (s) {
#define MM_CASE(S) case S: fmac_bc<S>(acc, b, m); break;
    MM_CASE(0) MM_CASE(1) MM_CASE(2) MM_CASE(3) MM_CASE(4) MM_CASE(5) MM_CASE(6) MM_CASE(7) MM_CASE(8) MM_CASE(9) MM_CASE(10)
    MM_CASE(11) MM_CASE(12) MM_CASE(13) MM_CASE(14)
    default: fmac_bc<15>(acc, b, m); break;
#undef MM_CASE
  }
}
__device__ __forceinline__ float mul_sel(int s, float b, float m) {
  switch (s) {
#define MM_CASE(S) case S: return mul_bc<S>(b, m);
    MM_CASE(0) MM_CASE(1) MM_CASE(2) MM_CASE(3) MM_CASE(4) MM_CASE(5) MM_CASE(6) MM_CASE(7) MM_CASE(8) MM_CASE(9) MM_CASE(10)
    MM_CASE(11) MM_CASE(12) MM_CASE(13) MM_CASE(14)
    default: return mul_bc<15>(b, m);
#undef MM_CASE
  }
}

template <bool SP>
__global__ __launch_bounds__(256) void scan_fwd_rows_kernel(const FwdParams p) {
  constexpr int CH = 16, kT = 64, kTS = kT + 4, QL = kT / 4, RPI = kWave / QL, NLD = CH / RPI;
  constexpr int WLDS = 2 * CH * kTS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb = blockDim.x >> 6;
  int blk = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) blk = (blk & 7) * (nblk >> 3) + (blk >> 3);      // XCD-aware numbering, as above
  const int gw = blk * wpb + wave;
  if (gw >= p.nwaves_total) return;
  const int cwv = gw % p.wpg;
  const int bk = gw / p.wpg;
  const int grp = bk % p.G, b = bk / p.G;
  const int ugrp = p.ug < p.G ? (int)((p.u_map >> (4 * grp)) & 15) : grp;
  const bool rev = grp < 32 && ((p.rev_mask >> grp) & 1);

  float* wl = smem + wave * WLDS;
  float* s_dl = wl;                      // [CH][kTS] delta'
  float* s_du = wl + CH * kTS;           // [CH][kTS] delta'*u, then y (in place)

  // ---- recurrence identity: row g = lane / 16 holds states 4g..4g+3, column c = lane % 16 is the channel
  const int g = lane >> 4, c = lane & 15;
  const int hc = cwv * CH + c;
  const bool cvalid = hc < p.H;
  const int d = grp * p.H + (cvalid ? hc : 0);
  v2f A01, A23, x01 = {0.f, 0.f}, x23 = {0.f, 0.f};
  {
    const float* Ad = p.A + (int64_t)d * kNState + 4 * g;
    A01 = (v2f){Ad[0] * kLog2e, Ad[1] * kLog2e};
    A23 = (v2f){Ad[2] * kLog2e, Ad[3] * kLog2e};
  }
  // ---- staging identity (u, delta, out): row r of a 4-row group, float4 column q — numerically (g, c) again
  const int r = lane / QL, q = lane % QL;
  const int hc0 = cwv * CH + r;
  const int d0 = grp * p.H + cwv * CH;
  const int d0u = ugrp * p.H + cwv * CH;
  const int nrw = min(CH, p.H - cwv * CH);
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
    rvalid[i] = hc0 + RPI * i < p.H;
    const int dd = grp * p.H + (rvalid[i] ? hc0 + RPI * i : 0);
    Dv[i] = p.D ? p.D[dd] : 0.f;
    bv[i] = p.bias ? p.bias[dd] : 0.f;
    uoff[i] = (int)((r + RPI * i) * p.u_sd) * 4;
    doff[i] = (int)((r + RPI * i) * p.d_sd) * 4;
    ooff[i] = ((r + RPI * i) * p.L) * 4;
  }
  int boff[4], coff[4];                  // this lane's four B / C rows (states 4g + j)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    boff[j] = (int)((4 * g + j) * p.B_sn) * 4;
    coff[j] = (int)((4 * g + j) * p.C_sn) * 4;
  }
  // B / C of the 16 steps that start at time tb: lane c takes step tb + c (memory position L-1-t for a reversed direction)
  auto load_bc = [&](int tb, float (&Bq)[4], float (&Cq)[4]) {
    const int t = tb + c;
    const bool ok = t < p.L;
    const int pos = (rev ? p.L - 1 - t : t) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Bq[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB, ok ? boff[j] + pos : kOOB, 0, 0));
      Cq[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rC, ok ? coff[j] + pos : kOOB, 0, 0));
    }
  };

  float4 pu[NLD], pd[NLD];
  auto issue_loads = [&](int t0) {
    const int t = t0 + 4 * q;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      pu[i] = load_quad<true, true>(ru, uoff[i], t, p.L, rev, rvalid[i]);
      pd[i] = load_quad<true, true>(rd, doff[i], t, p.L, rev, rvalid[i]);
    }
  };
  const int qc = rev ? QL - 1 - q : q;   // a reversed tile lies mirrored in LDS (quads in memory order), as above
  float4 yreg[NLD];
  auto store_tile = [&](int t0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) store_quad<true, true>(ro, ooff[i], t0 + 4 * q, p.L, rev, rvalid[i], yreg[i]);
  };
  // which of a group's four steps this row holds after the cross-row sum (see below): rows 0..3 -> steps 0, 2, 1, 3
  const int ystep = (g & 1) * 2 + (g >> 1);

  float Ba[4], Ca[4], Bb[4], Cb[4];      // two register sets: the block in use and the block in flight
  issue_loads(0);
  load_bc(0, Ba, Ca);
  for (int tile = 0; tile < p.ntiles; ++tile) {
    const int t0 = tile * kT;
    // ---- phase 1: registers -> LDS (delta' and delta'*u), keep D*u for the epilogue
    float4 uD[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int t = t0 + 4 * q;
      float4 dl, du;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float raw = f4get(pd[i], e) + bv[i];
        float v = SP ? softplus_f(raw) : raw;
        v = (rvalid[i] && t < p.L) ? v : 0.f;         // identity steps outside the sequence / channel range
        (&dl.x)[e] = v;
        (&du.x)[e] = v * f4get(pu[i], e);
      }
      uD[i] = make_float4(pu[i].x * Dv[i], pu[i].y * Dv[i], pu[i].z * Dv[i], pu[i].w * Dv[i]);
      const int off = (RPI * i + r) * kTS + 4 * qc;
      *reinterpret_cast<float4*>(s_dl + off) = dl;
      *reinterpret_cast<float4*>(s_du + off) = du;
    }
    if (tile > 0) store_tile(t0 - kT);                 // older than the loads issued next (see the kernel above)
    if (tile + 1 < p.ntiles) issue_loads(t0 + kT);

    // ---- phase 2: the recurrence, blocks of 16 steps = 4 groups of 4
    const int tlen = min(kT, p.L - t0);
    const int ngroups = (tlen + 3) >> 2;
    const int nblocks = (tlen + 15) >> 4;
    auto phase2 = [&](auto rvtag) {
      constexpr bool RV = decltype(rvtag)::value;
      struct Ops { float4 dl4, du4; };
      auto load_ops = [&](int tg) {
        Ops o;
        const int col = RV ? QL - 1 - tg : tg;
        o.dl4 = *reinterpret_cast<const float4*>(s_dl + c * kTS + 4 * col);
        o.du4 = *reinterpret_cast<const float4*>(s_du + c * kTS + 4 * col);
        return o;
      };
      auto at = [](const float4& v, int e) { return f4get(v, RV ? 3 - e : e); };
      // one group of 4 steps; S0 = index of its first step inside the 16-step block (compile time: it is the DPP lane select)
      auto group = [&](auto stag, const Ops& o, const float (&Bq)[4], const float (&Cq)[4], int tg) {
        constexpr int S0 = decltype(stag)::value;
        v2f a01[4], a23[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const v2f p01 = A01 * at(o.dl4, e), p23 = A23 * at(o.dl4, e);
          a01[e] = (v2f){__builtin_amdgcn_exp2f(p01.x), __builtin_amdgcn_exp2f(p01.y)};
          a23[e] = (v2f){__builtin_amdgcn_exp2f(p23.x), __builtin_amdgcn_exp2f(p23.y)};
        }
        __builtin_amdgcn_sched_barrier(0);       // all decay factors first: a v_exp_f32 result consumed at once stalls a lone wave
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float du = at(o.du4, e);
          const v2f t01 = a01[e] * x01, t23 = a23[e] * x23;
          float t0_ = t01.x, t1_ = t01.y, t2_ = t23.x, t3_ = t23.y;
          fmac_sel(S0 + e, t0_, Bq[0], du);
          fmac_sel(S0 + e, t1_, Bq[1], du);
          fmac_sel(S0 + e, t2_, Bq[2], du);
          fmac_sel(S0 + e, t3_, Bq[3], du);
          float yy = mul_sel(S0 + e, Cq[0], t0_);
          fmac_sel(S0 + e, yy, Cq[1], t1_);
          fmac_sel(S0 + e, yy, Cq[2], t2_);
          fmac_sel(S0 + e, yy, Cq[3], t3_);
          x01 = (v2f){t0_, t1_};
          x23 = (v2f){t2_, t3_};
          y[e] = yy;
        }
        // sum over the four rows: swap32 + add leaves (y0 | y1) resp. (y2 | y3) summed over row pairs in the two wave halves,
        // swap16 + add finishes both: rows 0..3 end with the sums of steps 0, 2, 1, 3
        const auto r01 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, y[0]), __builtin_bit_cast(unsigned, y[1]), false, false);
        const unsigned u0 = r01[0], u1 = r01[1];
        const float s01 = __builtin_bit_cast(float, u0) + __builtin_bit_cast(float, u1);
        const auto r23 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, y[2]), __builtin_bit_cast(unsigned, y[3]), false, false);
        const unsigned u2 = r23[0], u3 = r23[1];
        const float s23 = __builtin_bit_cast(float, u2) + __builtin_bit_cast(float, u3);
        const auto rr = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s01), __builtin_bit_cast(unsigned, s23), false, false);
        const unsigned w0 = rr[0], w1 = rr[1];
        const float yt = __builtin_bit_cast(float, w0) + __builtin_bit_cast(float, w1);
        const int tpos = 4 * tg + ystep;                                   // time step inside the tile
        s_du[c * kTS + (RV ? kT - 1 - tpos : tpos)] = yt;
      };
      auto checkpoint = [&](int blk16) {
        if (p.x_chk != nullptr && cvalid) {
          // (batch, chunk, dim, 16): the 16 states of the wave's 16 channels are one contiguous 1-KB run (see above)
          float* dst = p.x_chk + (((int64_t)b * p.nchk + (t0 >> 4) + blk16) * p.dim + d) * kNState + 4 * g;
          *reinterpret_cast<float4*>(dst) = make_float4(x01.x, x01.y, x23.x, x23.y);
        }
      };
      // four groups of one block: operand sets rotate by name (A, B, A, B), so opA holds the next block's first group at the end
      auto block16 = [&](int blk16, const float (&Bq)[4], const float (&Cq)[4], Ops& opA) {
        const int tg0 = 4 * blk16;
        Ops opB = load_ops(min(tg0 + 1, ngroups - 1));
        __builtin_amdgcn_sched_barrier(0);
        group(std::integral_constant<int, 0>{}, opA, Bq, Cq, tg0);
        __builtin_amdgcn_sched_barrier(0);
        if (tg0 + 1 < ngroups) {
          opA = load_ops(min(tg0 + 2, ngroups - 1));
          __builtin_amdgcn_sched_barrier(0);
          group(std::integral_constant<int, 4>{}, opB, Bq, Cq, tg0 + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (tg0 + 2 < ngroups) {
          opB = load_ops(min(tg0 + 3, ngroups - 1));
          __builtin_amdgcn_sched_barrier(0);
          group(std::integral_constant<int, 8>{}, opA, Bq, Cq, tg0 + 2);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (tg0 + 3 < ngroups) {
          opA = load_ops(min(tg0 + 4, ngroups - 1));
          __builtin_amdgcn_sched_barrier(0);
          group(std::integral_constant<int, 12>{}, opB, Bq, Cq, tg0 + 3);
          __builtin_amdgcn_sched_barrier(0);
        } else if (tg0 + 2 < ngroups) {
          opA = opB;
        }
        checkpoint(blk16);
      };
      Ops opA = load_ops(0);
      for (int blk16 = 0; blk16 < nblocks; blk16 += 2) {
        load_bc(t0 + 16 * (blk16 + 1), Bb, Cb);        // the next block's B / C fly while this block computes
        block16(blk16, Ba, Ca, opA);
        if (blk16 + 1 < nblocks) {
          load_bc(t0 + 16 * (blk16 + 2), Ba, Ca);      // (blk16 + 2 == 4: the next tile's first block)
          block16(blk16 + 1, Bb, Cb, opA);
        }
      }
    };
    if (rev) phase2(std::true_type{}); else phase2(std::false_type{});

    // ---- phase 3: y (+ D*u) LDS -> registers -> global
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const float4 y = *reinterpret_cast<const float4*>(s_du + (RPI * i + r) * kTS + 4 * qc);
      yreg[i] = make_float4(y.x + uD[i].x, y.y + uD[i].y, y.z + uD[i].z, y.w + uD[i].w);
    }
  }
  store_tile((p.ntiles - 1) * kT);
}


template <bool SP>
int launch_rows(const FwdParams& p, int nblocks, int wpb, hipStream_t stream) {
  const size_t lds = sizeof(float) * (size_t)wpb * (2 * 16 * (64 + 4));
  FwdParams q = p;
  q.ntiles = (p.L + 63) / 64;
  hipLaunchKernelGGL((scan_fwd_rows_kernel<SP>), dim3(nblocks), dim3(wpb * 64), lds, stream, q);
  return (int)hipGetLastError();
}

