// Micro-benchmark (diagnostic, not product): the inner recurrence of the forward selective scan for 16 channels x 16 states per
// wavefront, operands served from wave-private LDS only (no global traffic), in three forms, against waves per SIMD:
//   P  lane = (channel, 4-state group = lane % 4); B/C rows [position][16 states] in LDS, two ds_read_b128 per step and lane;
//      state pairs on v_pk_mul_f32 / v_pk_fma_f32; y summed over the quad by DPP, stored by one lane in four   (= scan_fwd.hip)
//   D  lane = (4-state group = lane / 16, channel = lane % 16); B/C of 16 steps in 8 VGPRs (lane c holds step c), read by the
//      FMAs through DPP row_newbcast — no LDS read of B/C in the loop; y summed over the four rows by three permlane swaps
//   D3 as D, with the two multiplies per state (A2*delta', a*x) packed in pairs (v_pk_mul_f32), the DPP FMAs on the pair halves
// Reports ns per time step and wavefront, and the time a 3136-step sequence (56x56 stage) would take at that rate.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fwd_loop tools/ubench/fwd_loop.hip && /tmp/fwd_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kTS = 68, kBCS = 20;
constexpr float kLog2e = 1.4426950408889634f;

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float f4get(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// ------------------------------------------------------------------------------------------------ P (the shipped form)
template <int kTile>      // 64: the long-sequence form; 32: the LEAN form (half the LDS per wave: 4 waves per SIMD fit)
__global__ __launch_bounds__(64) void loop_P(float* out, const float* in, int iters) {
  constexpr int QL = kTile / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_dl = smem; float* s_du = smem + 16 * kTS; float* s_bc = smem + 2 * 16 * kTS;
  const int lane = threadIdx.x;
  for (int i = lane; i < 2 * 16 * kTS; i += 64) smem[i] = 0.01f + 0.05f * in[i & 1023];
  for (int i = lane; i < 2 * kTile * kBCS; i += 64) s_bc[i] = in[(i * 7) & 1023] - 0.5f;
  __syncthreads();
  const int c = lane / 4, g = lane % 4;
  float A2[4], x[4];
  for (int j = 0; j < 4; ++j) { A2[j] = -(float)(g * 4 + j + 1) * kLog2e; x[j] = 0.f; }
  const float* sB = s_bc + g * 4; const float* sC = s_bc + kTile * kBCS + g * 4;
  struct Ops { float4 dl4, du4; float4 Bt[4], Ct[4]; };
  auto load_ops = [&](int tg) {
    Ops o;
    o.dl4 = *reinterpret_cast<const float4*>(s_dl + c * kTS + 4 * tg);
    o.du4 = *reinterpret_cast<const float4*>(s_du + c * kTS + 4 * tg);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = e * QL + tg;
      o.Bt[e] = *reinterpret_cast<const float4*>(sB + row * kBCS);
      o.Ct[e] = *reinterpret_cast<const float4*>(sC + row * kBCS);
    }
    return o;
  };
  float acc = 0.f;
  auto compute = [&](const Ops& o, int tg) {
    v2f a[4][2];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const v2f pw = (v2f){A2[2 * jj], A2[2 * jj + 1]} * f4get(o.dl4, e);
        a[e][jj] = (v2f){__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
      }
    __builtin_amdgcn_sched_barrier(0);
    float4 y4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float du = f4get(o.du4, e);
      v2f yy = {0.f, 0.f};
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        v2f xx = {x[2 * jj], x[2 * jj + 1]};
        const v2f Bp = jj ? (v2f){o.Bt[e].z, o.Bt[e].w} : (v2f){o.Bt[e].x, o.Bt[e].y};
        const v2f Cp = jj ? (v2f){o.Ct[e].z, o.Ct[e].w} : (v2f){o.Ct[e].x, o.Ct[e].y};
        xx = a[e][jj] * xx + Bp * du;
        yy = xx * Cp + yy;
        x[2 * jj] = xx.x; x[2 * jj + 1] = xx.y;
      }
      float y = yy.x + yy.y;
      y += dpp_f<0xB1>(y); y += dpp_f<0x4E>(y);
      (&y4.x)[e] = y;
    }
    if (g == 0) *reinterpret_cast<float4*>(s_du + c * kTS + 4 * tg) = y4;
    acc += y4.x;
  };
  for (int it = 0; it < iters; ++it) {
    Ops opA = load_ops(0);
    for (int tg = 0; tg < QL; tg += 2) {
      Ops opB = load_ops(tg + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(opA, tg);
      __builtin_amdgcn_sched_barrier(0);
      opA = load_ops(tg + 2 < QL ? tg + 2 : QL - 1);
      __builtin_amdgcn_sched_barrier(0);
      compute(opB, tg + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  out[blockIdx.x * 64 + lane] = acc + x[0] + x[1] + x[2] + x[3];
}

// ------------------------------------------------------------------------------------------------ D / D2
// LDS: dl, du [16][kTS]; y [16][kTS]; B, C as [n][t] rows of kTS floats (what a coalesced staging store writes without transposing)
template <int S> __device__ __forceinline__ float bcast(float v) { return dpp_f<0x150 + S>(v); }

template <int S> __device__ __forceinline__ void fmac_bc(float& acc, float b, float m) {     // acc += lane S of b's row * m
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(S));
}
template <int S> __device__ __forceinline__ float mul_bc(float b, float m) {
  float r;
  asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(b), "v"(m), "n"(S));
  return r;
}
// the step index is a constant after unrolling: the switch folds to one case
__device__ __forceinline__ void fmac_sel(int s, float& acc, float b, float m) {
  switch (s) {
#define CASE(S) case S: fmac_bc<S>(acc, b, m); break;
    CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
    default: fmac_bc<15>(acc, b, m); break;
#undef CASE
  }
}
__device__ __forceinline__ float mul_sel(int s, float b, float m) {
  switch (s) {
#define CASE(S) case S: return mul_bc<S>(b, m);
    CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
    default: return mul_bc<15>(b, m);
#undef CASE
  }
}

template <bool PACKED>
__global__ __launch_bounds__(64) void loop_D(float* out, const float* in, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // dl, du tiles as in the kernel (y overwrites du in place); the B / C registers are loaded from a small [16][64] table each
  // (in the kernel they come from global memory / L2 by buffer_load_dword: no LDS at all): 12.9 KB per wave, 12 waves per CU fit
  float* s_dl = smem; float* s_du = smem + 16 * kTS; float* s_y = s_du; float* s_b = smem + 2 * 16 * kTS;
  float* s_c = s_b + 16 * 16;
  const int lane = threadIdx.x;
  for (int i = lane; i < 2 * 16 * kTS; i += 64) smem[i] = 0.01f + 0.05f * in[i & 1023];
  for (int i = lane; i < 2 * 16 * 16; i += 64) s_b[i] = in[(i * 7) & 1023] - 0.5f;
  __syncthreads();
  const int g = lane / 16, c = lane % 16;
  float A2[4], x[4];
  for (int j = 0; j < 4; ++j) { A2[j] = -(float)(g * 4 + j + 1) * kLog2e; x[j] = 0.f; }
  float acc = 0.f;
  const int yrow = (g == 0 ? 0 : g == 1 ? 2 : g == 2 ? 1 : 3);   // which step of a 4-step group this row ends up holding
  auto load_bc = [&](int blk, float (&Bq)[4], float (&Cq)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Bq[j] = s_b[(4 * g + j) * 16 + ((c + blk) & 15)];
      Cq[j] = s_c[(4 * g + j) * 16 + ((c + 3 * blk) & 15)];
    }
  };
  auto block16 = [&](int blk, const float (&Bq)[4], const float (&Cq)[4]) {
    float4 dlA = *reinterpret_cast<const float4*>(s_dl + c * kTS + 16 * blk);
    float4 duA = *reinterpret_cast<const float4*>(s_du + c * kTS + 16 * blk);
    auto group = [&](auto gtag, const float4& dl4, const float4& du4) {
      constexpr int G4 = decltype(gtag)::value;
      float y[4];
      float a[4][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (PACKED) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const v2f pw = (v2f){A2[2 * jj], A2[2 * jj + 1]} * f4get(dl4, e);
            a[e][2 * jj] = __builtin_amdgcn_exp2f(pw.x); a[e][2 * jj + 1] = __builtin_amdgcn_exp2f(pw.y);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[e][j] = __builtin_amdgcn_exp2f(f4get(dl4, e) * A2[j]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float du = f4get(du4, e);
        if constexpr (!PACKED) {
          float yy;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float t = a[e][j] * x[j];
            fmac_sel(4 * G4 + e, t, Bq[j], du);                 // t += B[step][n] * du   (B read through DPP row_newbcast)
            x[j] = t;
            if (j == 0) yy = mul_sel(4 * G4 + e, Cq[j], t); else fmac_sel(4 * G4 + e, yy, Cq[j], t);
          }
          y[e] = yy;
        } else {
          // D3: the two multiplies of a state pair packed (v_pk_mul_f32: A2*delta' is done above, a*x here), the two FMAs that
          // read B / C through DPP on the halves of the pair registers
          float yy;
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            v2f t = (v2f){a[e][2 * jj], a[e][2 * jj + 1]} * (v2f){x[2 * jj], x[2 * jj + 1]};
            float t0 = t.x, t1 = t.y;
            fmac_sel(4 * G4 + e, t0, Bq[2 * jj], du);
            fmac_sel(4 * G4 + e, t1, Bq[2 * jj + 1], du);
            x[2 * jj] = t0; x[2 * jj + 1] = t1;
            if (jj == 0) yy = mul_sel(4 * G4 + e, Cq[0], t0); else fmac_sel(4 * G4 + e, yy, Cq[2], t0);
            fmac_sel(4 * G4 + e, yy, Cq[2 * jj + 1], t1);
          }
          y[e] = yy;
        }
      }
      // sum over the four rows: (y0,y1) and (y2,y3) by permlane32_swap + add, the two results by permlane16_swap + add
      const auto r01 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, y[0]), __builtin_bit_cast(unsigned, y[1]), false, false);
      const unsigned a0 = r01[0], a1 = r01[1];
      const float s01 = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
      const auto r23 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, y[2]), __builtin_bit_cast(unsigned, y[3]), false, false);
      const unsigned b0 = r23[0], b1 = r23[1];
      const float s23 = __builtin_bit_cast(float, b0) + __builtin_bit_cast(float, b1);
      const auto rr = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s01), __builtin_bit_cast(unsigned, s23), false, false);
      const unsigned c0 = rr[0], c1 = rr[1];
      const float yt = __builtin_bit_cast(float, c0) + __builtin_bit_cast(float, c1);   // row g: step {0,2,1,3}[g] of this group
      s_y[c * kTS + 16 * blk + 4 * G4 + yrow] = yt;
      acc += yt;
    };
    float4 dlB = *reinterpret_cast<const float4*>(s_dl + c * kTS + 16 * blk + 4);
    float4 duB = *reinterpret_cast<const float4*>(s_du + c * kTS + 16 * blk + 4);
    group(std::integral_constant<int, 0>{}, dlA, duA);
    dlA = *reinterpret_cast<const float4*>(s_dl + c * kTS + 16 * blk + 8);
    duA = *reinterpret_cast<const float4*>(s_du + c * kTS + 16 * blk + 8);
    group(std::integral_constant<int, 1>{}, dlB, duB);
    dlB = *reinterpret_cast<const float4*>(s_dl + c * kTS + 16 * blk + 12);
    duB = *reinterpret_cast<const float4*>(s_du + c * kTS + 16 * blk + 12);
    group(std::integral_constant<int, 2>{}, dlA, duA);
    group(std::integral_constant<int, 3>{}, dlB, duB);
  };
  for (int it = 0; it < iters; ++it) {
    float B0[4], C0[4], B1[4], C1[4];
    load_bc(0, B0, C0);
    load_bc(1, B1, C1);
    block16(0, B0, C0);
    load_bc(2, B0, C0);
    block16(1, B1, C1);
    load_bc(3, B1, C1);
    block16(2, B0, C0);
    block16(3, B1, C1);
  }
  out[blockIdx.x * 64 + lane] = acc + x[0] + x[1] + x[2] + x[3];
}


// ---- per-instruction rates: R independent chains of one instruction form, per waves/SIMD
enum { I_FMA, I_EXP, I_PKFMA, I_PKMUL, I_FMAC_DPP, I_MIX, I_SWAP32, I_MIXPK };
template <int KIND>
__global__ __launch_bounds__(64) void rate_k(float* out, float seed, int iters) {
  float a[8]; v2f pa[8];
  for (int j = 0; j < 8; ++j) { a[j] = seed * (j + 1) + threadIdx.x * 1e-6f; pa[j] = (v2f){a[j], a[j] * 0.5f}; }
  const float b = 0.999f, c = 1e-3f; const v2f pb = {0.999f, 0.998f}, pc = {1e-3f, 2e-3f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (KIND == I_FMA) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
      } else if constexpr (KIND == I_EXP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(a[j]));
      } else if constexpr (KIND == I_PKFMA) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa[j]) : "v"(pb), "v"(pc));
      } else if constexpr (KIND == I_PKMUL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[j]) : "v"(pb));
      } else if constexpr (KIND == I_FMAC_DPP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(b), "v"(c));
      } else if constexpr (KIND == I_MIX) {      // the unpacked recurrence mix: 2 exp + 2 mul + 2 mul + 2 fma + 2 fma per 2 states
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          asm volatile("v_exp_f32 %0, %0" : "+v"(a[j]));
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j + 1]) : "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j + 1]) : "v"(b), "v"(c));
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j + 1]) : "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j + 1]) : "v"(b), "v"(c));
        }
      } else if constexpr (KIND == I_MIXPK) {    // packed mix: 2 exp + pk_mul + pk_mul + pk_fma + pk_fma per 2 states
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          asm volatile("v_exp_f32 %0, %0" : "+v"(a[j]));
          asm volatile("v_exp_f32 %0, %0" : "+v"(a[j + 1]));
          asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[j]) : "v"(pb));
          asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pa[j + 1]) : "v"(pb));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa[j]) : "v"(pb), "v"(pc));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa[j + 1]) : "v"(pb), "v"(pc));
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; j += 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[j]), "+v"(a[j + 1]));
      }
    }
  }
  float s = 0; for (int j = 0; j < 8; ++j) s += a[j] + pa[j].x + pa[j].y;
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int KIND>
void rate(const char* name, int instr_per_iter, float* d) {
  const int iters = 2048;
  for (int wq = 4; wq <= 16; wq *= 2) {
    const int blocks = 256 * wq;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 0.5f, iters);
    (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(rate_k<KIND>, dim3(blocks), dim3(64), 0, 0, d, 0.5f, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("rate %-22s waves/SIMD %d: %6.2f ns per instruction per SIMD (%5.2f cycles at 2.1 GHz)\n", name, wq / 4,
           ms * 1e6 / ((double)iters * instr_per_iter * (wq / 4)), ms * 1e6 / ((double)iters * instr_per_iter * (wq / 4)) * 2.1);
  }
}
void instr_rates(float* d) {
  rate<I_FMA>("v_fma_f32", 32, d);
  rate<I_EXP>("v_exp_f32", 32, d);
  rate<I_PKFMA>("v_pk_fma_f32", 32, d);
  rate<I_PKMUL>("v_pk_mul_f32", 32, d);
  rate<I_FMAC_DPP>("v_fmac_f32_dpp newbcast", 32, d);
  rate<I_MIX>("mix 1exp+2mul+2fma", 80, d);
  rate<I_MIXPK>("mix 2exp+2pkmul+2pkfma", 96, d);
  rate<I_SWAP32>("v_permlane32_swap", 16, d);
}

template <typename K>
void run(const char* name, K kern, size_t lds, float* d_out, const float* d_in, int tile = 64) {
  const int iters = 200 * 64 / tile;      // x tile steps = 12800 steps
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int wq = 4; wq <= 16; wq += (wq < 8 ? 2 : 4)) {     // waves per SIMD x 4: 1, 1.5, 2, 3, 4
    const int blocks = 256 * wq;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, d_out, d_in, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, d_out, d_in, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double ns_step = ms * 1e6 / 12800.0;
    // a launch of W = blocks wavefronts over 1024 SIMDs; all resident at once here, so the kernel time is one wave's serial time
    printf("%-3s LDS %5.1f KB waves/SIMD %.1f: %7.3f ms  %6.1f ns per step and wave, %5.1f ns per wave-step of the SIMD | chip %.2f T state-steps/s\n",
           name, lds / 1024.0, wq / 4.0, ms, ns_step, ns_step / (wq / 4.0), blocks * 256.0 * 12800 / ms / 1e9);
  }
}

int main() {
  float *d_out, *d_in;
  (void)hipMalloc(&d_out, 256 * 16 * 64 * sizeof(float));
  (void)hipMalloc(&d_in, 1024 * sizeof(float));
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)(rand() & 0xffff) / 65536.f;
  (void)hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
  run("P64", loop_P<64>, sizeof(float) * (2 * 16 * kTS + 2 * 64 * kBCS), d_out, d_in, 64);
  run("P32", loop_P<32>, sizeof(float) * (2 * 16 * kTS + 2 * 32 * kBCS), d_out, d_in, 32);
  run("D", loop_D<false>, sizeof(float) * (2 * 16 * kTS + 2 * 16 * 16), d_out, d_in);
  run("D3", loop_D<true>, sizeof(float) * (2 * 16 * kTS + 2 * 16 * 16), d_out, d_in);
  instr_rates(d_out);
  return 0;
}
