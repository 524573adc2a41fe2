// Shared device helpers for the gfx950 kernels (wave64 only; no portability layer on purpose).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mm {

constexpr int kWave = 64;
constexpr int kNState = 16;      // d_state of every MedMamba block (MedMamba.py:329, 457)
constexpr int kTile = 64;        // timesteps staged through LDS per tile
constexpr int kTileStride = 68;  // LDS row stride in floats: +16 B pad -> conflict-free b128 rows
constexpr int kChunk = 16;       // checkpoint interval (steps) of x_chk
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// DPP controls (gfx9 encoding)
constexpr int DPP_QUAD_XOR1 = 0xB1;    // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;    // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_ROR8 = 0x128;

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Butterfly sum over the WIDTH (1,2,4,8,16) consecutive lanes that share lane/WIDTH; every lane gets the sum.
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (WIDTH >= 2) v += dpp_f<DPP_QUAD_XOR1>(v);
  if constexpr (WIDTH >= 4) v += dpp_f<DPP_QUAD_XOR2>(v);
  if constexpr (WIDTH >= 8) v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
  if constexpr (WIDTH >= 16) v += dpp_f<DPP_ROW_MIRROR>(v);
  return v;
}

// softplus with F.softplus semantics (beta 1, threshold 20: temp.py:63-64), built from v_exp/v_log.
// e = exp(x); for e < 2^-12 the series log1p(e) = e - e^2/2 is exact to fp32 and avoids 1+e rounding.
__device__ __forceinline__ float softplus_f(float x) {
  const float e = __builtin_amdgcn_exp2f(x * kLog2e);
  const float big = __builtin_amdgcn_logf(1.0f + e) * kLn2;   // v_log_f32 is log2
  const float small = e - 0.5f * e * e;
  float r = e < 2.44140625e-4f ? small : big;
  return x > 20.0f ? x : r;
}

__device__ __forceinline__ float sigmoid_f(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * kLog2e));
}

// (quotient, remainder) of a loop index that advances by a fixed step: one division before the loop instead of one per
// element (an integer division by a runtime value is ~25 VALU instructions; these kernels do ~60 of real work per element)
struct DivMod {
  int q, r, dq, dr, d;
  __device__ __forceinline__ DivMod(int start, int step, int div) : q(start / div), r(start % div), dq(step / div), dr(step % div), d(div) {}
  __device__ __forceinline__ void next() { q += dq; r += dr; if (r >= d) { r -= d; ++q; } }
};

// ---- bounds-checked buffer access (tails read 0 / drop stores; no divergent control flow) -------------
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
// NOTE: never __builtin_bit_cast a single ext-vector ELEMENT (v.x): hipcc 7.2 miscompiles it to element 0;
// bit_cast the whole vector and then take components.
constexpr int kOOB = 0x7fffffff;   // voffset that always fails the range check

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, int64_t bytes) {
  // a wave that lies completely outside the tensor gets an EMPTY descriptor (every access out of range);
  // a negative size would wrap to 4 GiB and turn the kOOB offset into a real address
  const int64_t capped = bytes <= 0 ? 0 : (bytes > 0x7ffffff0ll ? 0x7ffffff0ll : bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)capped, 0x00020000);
}

// Time-ordered values of time slots t..t+3 of the row that starts at byte offset `row` of descriptor r.
// rev: slot t lives at memory position L-1-t (the row is walked backwards; nothing is ever flipped in memory).
// VEC (L % 4 == 0, 16-B aligned rows): one dwordx4, a quad is entirely inside or outside the row.
// MEMORDER (VEC only): return the quad as it lies in memory even for a reversed row (components NOT swapped) — for callers
// that keep a reversed tile in memory order and walk it backwards themselves (4 v_cndmask per quad saved).
template <bool VEC, bool MEMORDER = false>
__device__ __forceinline__ float4 load_quad(rsrc_t r, int row, int t, int L, bool rev, bool rowok) {
  if constexpr (VEC) {
    const bool ok = rowok && t < L;
    const int pos = rev ? L - 4 - t : t;
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? row + pos * 4 : kOOB, 0, 0);
    const v4f f = __builtin_bit_cast(v4f, v);
    if constexpr (MEMORDER) return make_float4(f.x, f.y, f.z, f.w);
    return rev ? make_float4(f.w, f.z, f.y, f.x) : make_float4(f.x, f.y, f.z, f.w);
  } else {
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int te = t + e;
      const int pos = rev ? L - 1 - te : te;
      const unsigned a = __builtin_amdgcn_raw_buffer_load_b32(r, (rowok && te < L) ? row + pos * 4 : kOOB, 0, 0);
      o[e] = __builtin_bit_cast(float, a);
    }
    return make_float4(o[0], o[1], o[2], o[3]);
  }
}

template <bool VEC, bool MEMORDER = false>
__device__ __forceinline__ void store_quad(rsrc_t r, int row, int t, int L, bool rev, bool rowok, float4 v) {
  if constexpr (VEC) {
    const bool ok = rowok && t < L;
    const int pos = rev ? L - 4 - t : t;
    const v4f f = (!MEMORDER && rev) ? (v4f){v.w, v.z, v.y, v.x} : (v4f){v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, f), r, ok ? row + pos * 4 : kOOB, 0, 0);
  } else {
    const float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int te = t + e;
      const int pos = rev ? L - 1 - te : te;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[e]), r,
                                            (rowok && te < L) ? row + pos * 4 : kOOB, 0, 0);
    }
  }
}


}  // namespace mm
